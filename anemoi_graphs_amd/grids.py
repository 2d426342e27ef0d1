"""Synthetic lat/lon grids of the shapes the benchmark and the parity tests are quoted on.

All generators return float32 ``[N, 2]`` arrays of (latitude, longitude) in **radians**,
converted from degrees with ``np.deg2rad`` exactly as the reference's node builders do
(``/root/reference/src/anemoi/graphs/nodes/builders/base.py:99-101``).
"""
from __future__ import annotations

import numpy as np


def gauss_legendre_latitudes_deg(n_lat: int) -> np.ndarray:
    """The ``n_lat`` Gaussian latitudes in degrees, north to south (roots of P_{n_lat})."""
    n = int(n_lat)
    k = np.arange(1, n // 2 + 1, dtype=np.float64)
    # Tricomi initial guess, then Newton on P_n(x) via the three-term recurrence
    x = np.cos(np.pi * (4.0 * k - 1.0) / (4.0 * n + 2.0)) * (1.0 - 1.0 / (8.0 * n * n) + 1.0 / (8.0 * n**3))
    for _ in range(100):
        p0 = np.ones_like(x)
        p1 = x.copy()
        for j in range(2, n + 1):
            p0, p1 = p1, ((2.0 * j - 1.0) * x * p1 - (j - 1.0) * p0) / j
        dp = n * (x * p1 - p0) / (x * x - 1.0)
        dx = p1 / dp
        x = x - dx
        if np.max(np.abs(dx)) < 1e-15:
            break
    lat_north = np.rad2deg(np.arcsin(x))
    if n % 2:
        return np.concatenate([lat_north, [0.0], -lat_north[::-1]])
    return np.concatenate([lat_north, -lat_north[::-1]])


def octahedral_grid_deg(n: int) -> tuple[np.ndarray, np.ndarray]:
    """Octahedral reduced Gaussian grid O<n>: 2n latitude rows; the i-th row from each pole has
    4i+16 equally spaced longitudes starting at 0 degrees.  4n^2+36n points, north to south."""
    lats = gauss_legendre_latitudes_deg(2 * n)
    counts = np.concatenate([4 * np.arange(1, n + 1) + 16, (4 * np.arange(1, n + 1) + 16)[::-1]])
    lat_out = np.repeat(lats, counts)
    lon_out = np.concatenate([np.arange(c, dtype=np.float64) * (360.0 / c) for c in counts])
    return lat_out, lon_out


def regular_grid_deg(dlat: float, dlon: float | None = None) -> tuple[np.ndarray, np.ndarray]:
    """Regular lat/lon grid, latitude 90 -> -90 (poles included), longitude 0 -> 360-dlon, lat-major."""
    dlon = dlat if dlon is None else dlon
    nlat = int(round(180.0 / dlat)) + 1
    nlon = int(round(360.0 / dlon))
    lats = 90.0 - np.arange(nlat, dtype=np.float64) * dlat
    lons = np.arange(nlon, dtype=np.float64) * dlon
    return np.repeat(lats, nlon), np.tile(lons, nlat)


def to_radians_f32(lat_deg: np.ndarray, lon_deg: np.ndarray) -> np.ndarray:
    """``BaseNodeBuilder.reshape_coords`` (nodes/builders/base.py:99-101): stack, deg2rad, float32."""
    coords = np.stack([lat_deg, lon_deg], axis=-1).reshape((-1, 2))
    return np.deg2rad(coords).astype(np.float32)


def octahedral(n: int) -> np.ndarray:
    return to_radians_f32(*octahedral_grid_deg(n))


def regular(dlat: float, dlon: float | None = None) -> np.ndarray:
    return to_radians_f32(*regular_grid_deg(dlat, dlon))


def random_sphere(n: int, seed: int = 0) -> np.ndarray:
    """Uniform random points on the sphere (tie-free for parity runs)."""
    rng = np.random.default_rng(seed)
    lat = np.arcsin(rng.uniform(-1.0, 1.0, n))
    lon = rng.uniform(0.0, 2.0 * np.pi, n)
    return np.stack([lat, lon], axis=1).astype(np.float32)


def fibonacci_sphere(n: int) -> np.ndarray:
    """Quasi-uniform spiral point set (stand-in for an icosphere of ``n`` vertices)."""
    i = np.arange(n, dtype=np.float64) + 0.5
    lat = np.arcsin(1.0 - 2.0 * i / n)
    lon = np.mod(np.pi * (1.0 + 5.0**0.5) * i, 2.0 * np.pi)
    return np.stack([lat, lon], axis=1).astype(np.float32)


def named(name: str) -> np.ndarray:
    """``o96`` / ``o1280`` / ``era5`` (0.25 deg) / ``reg<deg>`` / ``rand<N>`` / ``fib<N>``."""
    name = name.lower()
    if name == "era5":
        return regular(0.25)
    if name.startswith("reg"):
        return regular(float(name[3:]))
    if name.startswith("rand"):
        return random_sphere(int(name[4:]))
    if name.startswith("fib"):
        return fibonacci_sphere(int(name[3:]))
    if name.startswith("o"):
        return octahedral(int(name[1:]))
    raise ValueError(f"unknown grid {name!r}")


# ---------------------------------------------------------------------------------------------------
# Stretched synthetic grid: a coarse global background with fine refinement discs
# ---------------------------------------------------------------------------------------------------
EARTH_RADIUS_KM = 6371.0


class StretchedGrid:
    """A global grid of ``coarse_km`` spacing with the inside of one or more discs refined to ``fine_km``
    (the shape of a stretched-grid dataset: a regional high-resolution area cut into a global one).

    The points lie on rows of constant latitude, ordered north -> south, so the grid can be cut into
    latitude bands by node id.  It is described by a small host-side table of *segments* -- arcs of a row
    with equally spaced longitudes -- and any range of node ids is generated on the device from that table,
    so no host array of the points ever exists and every rank of a sharded build makes just its own band:

    * fine rows, ``fine_km`` apart: inside every disc, the arc of the row that lies in the disc;
    * coarse rows, ``coarse_km`` apart and shifted by half a fine row so that they never coincide with one:
      the whole circle of latitude, without the arcs inside the discs.

    Longitudes are displaced by up to ``jitter`` of the segment's spacing with a counter-based generator (a
    64-bit mix of seed and node id): every point depends on its id alone, whoever generates it.

    ``discs``: (centre latitude, centre longitude, radius in km) in degrees / km; they must not overlap.
    """

    def __init__(self, coarse_km: float = 9.0, fine_km: float = 1.0,
                 discs: tuple[tuple[float, float, float], ...] = ((45.0, 10.0, 1500.0),), seed: int = 0,
                 jitter: float = 0.25) -> None:
        assert coarse_km >= fine_km > 0.0 and 0.0 <= jitter < 0.5
        self.coarse_km, self.fine_km, self.seed, self.jitter = float(coarse_km), float(fine_km), int(seed), float(jitter)
        self.discs = tuple((np.deg2rad(a), np.deg2rad(b) % (2 * np.pi), r / EARTH_RADIUS_KM) for a, b, r in discs)
        for i, (a1, b1, r1) in enumerate(self.discs):
            for a2, b2, r2 in self.discs[i + 1:]:
                cosd = np.sin(a1) * np.sin(a2) + np.cos(a1) * np.cos(a2) * np.cos(b1 - b2)
                assert np.arccos(np.clip(cosd, -1, 1)) > r1 + r2, "refinement discs must not overlap"
        lat, lon0, dlon, count, fine = self._segments()
        order = np.argsort(-lat, kind="stable")
        self.seg_lat, self.seg_lon0, self.seg_dlon = lat[order], lon0[order], dlon[order]
        self.seg_count, self.seg_fine = count[order], fine[order]
        self.seg_start = np.concatenate([[0], np.cumsum(self.seg_count)]).astype(np.int64)
        self._device_tables = {}

    def __len__(self) -> int:
        return int(self.seg_start[-1])

    # -- the table ---------------------------------------------------------------------------------
    def _half_width(self, lat: np.ndarray, disc) -> np.ndarray:
        """Half of the longitude interval a disc cuts out of the circles of latitude ``lat`` (negative: none)."""
        lat0, _, radius = disc
        with np.errstate(invalid="ignore", divide="ignore"):
            c = (np.cos(radius) - np.sin(lat) * np.sin(lat0)) / (np.cos(lat) * np.cos(lat0))
        half = np.arccos(np.clip(c, -1.0, 1.0))
        half = np.where(c >= 1.0, -1.0, half)  # the row misses the disc
        return np.where(np.abs(lat - lat0) >= radius, -1.0, half)

    def _segments(self):
        lat, lon0, dlon, count, fine = [], [], [], [], []
        step_f, step_c = self.fine_km / EARTH_RADIUS_KM, self.coarse_km / EARTH_RADIUS_KM
        rows_f = np.pi / 2 - (np.arange(int(round(np.pi / step_f))) + 0.5) * step_f
        for disc in self.discs:
            half = self._half_width(rows_f, disc)
            for phi, h in zip(rows_f[half >= 0.0], half[half >= 0.0]):
                spacing = step_f / max(np.cos(phi), 1e-12)
                if h >= np.pi - 0.5 * spacing:  # the whole circle
                    n = max(1, int(round(2 * np.pi / spacing)))
                    lat.append(phi); lon0.append(0.0); dlon.append(2 * np.pi / n); count.append(n); fine.append(True)
                    continue
                n = int(np.floor(2 * h / spacing)) + 1
                lat.append(phi); lon0.append(disc[1] - 0.5 * (n - 1) * spacing); dlon.append(spacing); count.append(n)
                fine.append(True)
        rows_c = np.pi / 2 - (np.arange(int(round(np.pi / step_c))) + 0.5) * step_c - 0.5 * step_f
        rows_c = rows_c[rows_c > -np.pi / 2]
        for i, phi in enumerate(rows_c):
            n = max(1, int(round(2 * np.pi * np.cos(phi) / step_c)))
            spacing = 2 * np.pi / n
            lon = (np.arange(n) + 0.5 * (i % 2)) * spacing
            keep = np.ones(n, dtype=bool)
            for disc in self.discs:
                h = float(self._half_width(np.array([phi]), disc)[0])
                if h >= 0.0:
                    # (a margin of half a fine spacing keeps coarse points off the rim of the disc)
                    d = np.abs((lon - disc[1] + np.pi) % (2 * np.pi) - np.pi)
                    keep &= d > h + 0.5 * step_f / max(np.cos(phi), 1e-12)
            if keep.all():
                lat.append(phi); lon0.append(lon[0]); dlon.append(spacing); count.append(n); fine.append(False)
                continue
            if not keep.any():
                continue
            # circular runs of kept points: start at a kept point that follows a removed one
            starts = np.where(keep & ~np.roll(keep, 1))[0]
            for s in starts:
                length = 0
                while keep[(s + length) % n] and length < n:
                    length += 1
                lat.append(phi); lon0.append(lon[s]); dlon.append(spacing); count.append(length); fine.append(False)
        return (np.array(lat), np.array(lon0), np.array(dlon), np.array(count, dtype=np.int64), np.array(fine, dtype=bool))

    # -- the points --------------------------------------------------------------------------------
    def _tables(self, device):
        import torch

        key = str(device)
        if key not in self._device_tables:
            self._device_tables[key] = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in (
                self.seg_start, self.seg_lat, self.seg_lon0, self.seg_dlon, self.seg_fine))
        return self._device_tables[key]

    @staticmethod
    def _uniform(ids, seed: int):
        """Counter-based uniform numbers in [-0.5, 0.5): the murmur3 64-bit finaliser of (id + seed * odd constant)."""
        import torch

        def shr(v, bits):  # logical shift of an int64 tensor
            return torch.bitwise_right_shift(v, bits) & ((1 << (64 - bits)) - 1)

        h = ids + (seed * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019) % (1 << 63)
        h = (h ^ shr(h, 33)) * -0xAE502812AA7333  # 0xff51afd7ed558ccd as a signed 64-bit integer
        h = (h ^ shr(h, 33)) * -0x3B314601E57A13AD  # 0xc4ceb9fe1a85ec53
        h = h ^ shr(h, 33)
        return shr(h, 40).to(torch.float64) * (1.0 / (1 << 24)) - 0.5

    def generate(self, lo: int = 0, hi: int | None = None, device=None, chunk: int = 1 << 25, return_fine: bool = False):
        """float32 [hi - lo, 2] (latitude, longitude) radians of the nodes lo .. hi-1 on ``device`` (and, with
        ``return_fine``, the bool mask of the nodes that belong to a refinement disc)."""
        import torch

        hi = len(self) if hi is None else hi
        assert 0 <= lo <= hi <= len(self)
        device = torch.device("cpu") if device is None else device
        start, lat, lon0, dlon, fine = self._tables(device)
        out = torch.empty((hi - lo, 2), dtype=torch.float32, device=device)
        mask = torch.empty(hi - lo, dtype=torch.bool, device=device) if return_fine else None
        for a in range(lo, hi, chunk):
            b = min(hi, a + chunk)
            ids = torch.arange(a, b, dtype=torch.int64, device=device)
            seg = torch.searchsorted(start, ids, right=True) - 1
            pos = (ids - start[seg]).to(torch.float64) + self.jitter * 2.0 * self._uniform(ids, self.seed)
            out[a - lo:b - lo, 0] = lat[seg].to(torch.float32)
            out[a - lo:b - lo, 1] = torch.remainder(lon0[seg] + pos * dlon[seg], 2.0 * np.pi).to(torch.float32)
            if return_fine:
                mask[a - lo:b - lo] = fine[seg]
        # (a float32 longitude may round up to 2 pi itself; keep it inside [0, 2 pi) like every other grid here)
        out[:, 1].clamp_(max=float(np.nextafter(np.float32(2 * np.pi), np.float32(0))))
        return (out, mask) if return_fine else out


def stretched_for_nodes(n_nodes: float, coarse_km: float = 9.0, fine_km: float = 1.0, centre=(45.0, 10.0),
                        seed: int = 0) -> StretchedGrid:
    """A stretched grid of about ``n_nodes`` nodes: the radius of the one refinement disc follows from the count
    (background 4 pi R^2 / coarse^2 nodes, disc area / fine^2 nodes inside)."""
    sphere = 4.0 * np.pi * EARTH_RADIUS_KM**2
    background = sphere / coarse_km**2
    frac = (n_nodes - background) / (sphere / fine_km**2 - background)  # share of the sphere that is refined
    frac = min(max(frac, 1.0e-6), 0.98)
    radius_km = np.arccos(1.0 - 2.0 * frac) * EARTH_RADIUS_KM
    return StretchedGrid(coarse_km, fine_km, ((centre[0], centre[1], radius_km),), seed)

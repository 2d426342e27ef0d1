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

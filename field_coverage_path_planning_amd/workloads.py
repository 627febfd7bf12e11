"""Synthetic inputs of the BASELINE.json configurations, shared by bench.py, the full-size GPU tests and tools/.

Every generator is a pure function of its seed (numpy PCG64) and returns plain Python / numpy data: (length, width) pairs or
vertex lists -- no device objects -- so the same inputs can be handed to the HIP engine (engine.FieldSpec) and to the CPU
oracle (oracle.make_field).  Nothing here reads the reference tree.
"""
import numpy as np

CFG1_LH = (500.0, 200.0)        # BASELINE.json configs[0]: the reference's own 500 x 200 m rectangle (README_en.md:199-215)


def cfg1_batch(n_fields=4096):
    """n copies of the 500 x 200 m field the metric is quoted on -> (n, 2) array of (L, H)."""
    return np.tile(np.array(CFG1_LH, dtype=np.float64), (int(n_fields), 1))


def cfg2_rectangles(n_fields=1024, seed=1024):
    """BASELINE.json configs[1]: random rectangles, edges U[100, 1000) m -> (n, 2) array of (L, H)."""
    return np.random.default_rng(seed).uniform(100.0, 1000.0, size=(int(n_fields), 2))


def cfg3_field(seed=32):
    """BASELINE.json configs[2]: one 5000 x 2000 m rectangle with 32 convex eight-gon obstacles, centres on a jittered 8 x 4 grid,
    circum-radius U[10, 40) m -> ((L, H), [polygon vertex lists])."""
    rng = np.random.default_rng(seed)
    obstacles = []
    for gy in range(4):
        for gx in range(8):
            cx = (gx + 0.5) * 5000 / 8 + rng.uniform(-100, 100)
            cy = (gy + 0.5) * 2000 / 4 + rng.uniform(-100, 100)
            r = rng.uniform(10, 40)
            obstacles.append([(float(cx + r * np.cos(a)), float(cy + r * np.sin(a))) for a in np.arange(8) * np.pi / 4])
    return (5000.0, 2000.0), obstacles


def cfg4_ga(n_nodes=128, population=4096, seed=128, pop_seed=4096):
    """BASELINE.json configs[3]: Euclidean distance matrix of n points U[0, 1000)^2 and a population of random tours
    -> (D float64 (n, n), routes int32 (population, n))."""
    pts = np.random.default_rng(seed).uniform(0, 1000, size=(int(n_nodes), 2))
    D = np.sqrt(((pts[:, None] - pts[None]) ** 2).sum(-1))
    rng = np.random.default_rng(pop_seed)
    routes = np.stack([rng.permutation(int(n_nodes)) for _ in range(int(population))]).astype(np.int32)
    return D, routes


def cfg5_parallelograms(n_fields=65536, seed=65536):
    """BASELINE.json configs[4]: parallelograms, base / height U[100, 1000) m, interior angle U[60, 120) degrees, rotated by
    U[-pi/4, pi/4), first vertex at the origin -> (n, 4, 2) array of vertices."""
    rng = np.random.default_rng(seed)
    out = np.empty((int(n_fields), 4, 2), dtype=np.float64)
    for k in range(int(n_fields)):
        base, height = rng.uniform(100, 1000, 2)
        ang, rot = np.radians(rng.uniform(60, 120)), rng.uniform(-np.pi / 4, np.pi / 4)
        sx = height / np.tan(ang)
        v = np.array([[0, 0], [base, 0], [base + sx, height], [sx, height]])
        out[k] = v @ np.array([[np.cos(rot), np.sin(rot)], [-np.sin(rot), np.cos(rot)]])
    return out


def specs_from_lh(E, LH):
    return [E.FieldSpec(field_length=float(a), field_width=float(b)) for a, b in LH]


def specs_from_vertices(E, V):
    return [E.FieldSpec(field_vertices=[(float(a), float(b)) for a, b in q]) for q in V]

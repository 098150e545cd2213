"""Mints tests/golden/terrain_ref.npz: height samples and tile origins produced by the REFERENCE's own `Terrain` class
(isaacgymenvs/utils/terrain.py run from /root/reference through oracle/ref_harness.py) for the configurations of
tests/test_terrain.py.  Test infrastructure; needs /root/reference, so it only runs in the build container.
usage: python oracle/make_terrain_goldens.py"""
import os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_harness

# (name, seed, overrides) -- the reference's rough-slope tile calls scipy's removed interp2d, so its proportion is 0
CASES = [
    ("curriculum_slopes_stairs", 3, dict(mesh_type="heightfield", curriculum=True, num_rows=4, num_cols=6, border_size=2,
                                         terrain_proportions=[0.4, 0.0, 0.3, 0.3, 0.0])),
    ("random_with_obstacles", 11, dict(mesh_type="heightfield", curriculum=False, num_rows=3, num_cols=4, border_size=2,
                                       terrain_proportions=[0.2, 0.0, 0.2, 0.2, 0.4])),
    ("stones_gap_pit", 5, dict(mesh_type="trimesh", curriculum=True, num_rows=3, num_cols=6, border_size=1,
                               terrain_proportions=[0.0, 0.0, 0.0, 0.0, 0.1, 0.3, 0.3])),
]


def main():
    ref = ref_harness.load_reference(lambda: None)
    ref_terrain = sys.modules["isaacgymenvs.utils.terrain"]
    ref_cfg = sys.modules["isaacgymenvs.cfg.terrain.terrain_cfg"]
    out = {}
    for name, seed, ov in CASES:
        cfg = ref_cfg.TerrainCfg()
        for k, v in ov.items():
            setattr(cfg, k, v)
        np.random.seed(seed)
        t = ref_terrain.Terrain(cfg, 64)
        out[name + "/heightsamples"] = np.asarray(t.heightsamples, dtype=np.int16)
        out[name + "/env_origins"] = np.asarray(t.env_origins, dtype=np.float64)
        if cfg.mesh_type == "trimesh":       # the mesh is large: keep digests and a few rows
            import hashlib
            out[name + "/vertices_sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(t.vertices).tobytes()).digest(), dtype=np.uint8)
            out[name + "/triangles_sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(t.triangles).tobytes()).digest(), dtype=np.uint8)
            out[name + "/vertices_shape"] = np.asarray(t.vertices.shape); out[name + "/triangles_shape"] = np.asarray(t.triangles.shape)
            out[name + "/vertices_dtype"] = np.asarray(str(t.vertices.dtype)); out[name + "/triangles_dtype"] = np.asarray(str(t.triangles.dtype))
        print(name, t.heightsamples.shape, int(t.heightsamples.min()), int(t.heightsamples.max()))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "terrain_ref.npz"), **out)


if __name__ == "__main__":
    main()

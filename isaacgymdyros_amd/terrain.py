"""Height-field terrain for DyrosDynamicWalk.

Mirrors what the reference builds when `TerrainCfg.mesh_type` is 'heightfield' or 'trimesh'
(reference: isaacgymenvs/cfg/terrain/terrain_cfg.py:1-22, isaacgymenvs/utils/terrain.py:40-165, the sub-terrain
shapes of isaacgym/terrain_utils.py:17-285): a grid of `num_rows` (difficulty levels) x `num_cols` (terrain types)
tiles of `terrain_length` x `terrain_width` metres inside a flat border, stored as int16 samples
(`height = sample * vertical_scale`, sample spacing `horizontal_scale`), plus one spawn origin per tile.

The generator is deterministic for a seed and draws from `numpy.random.RandomState(seed)` in the same order as the
reference draws from the global numpy generator, so `Terrain(cfg, n, seed=s)` reproduces the reference's samples
after `np.random.seed(s)` -- checked bit for bit in tests/test_terrain.py for every tile type whose reference code
still runs on this image (`random_uniform_terrain` calls scipy's removed `interp2d`; that type is held to its
specification, bilinear interpolation of a coarse random grid, instead).

The physics kernel consumes `heightsamples` as a device tensor: bilinear height and gradient under every contact
point (csrc/dw_physics.h, oracle/dw_physics.c).  A 'trimesh' terrain is simulated on the same samples -- the
reference's triangle mesh is that height field with near-vertical faces straightened -- and 'plane' needs none.
"""
from __future__ import annotations

import numpy as np


class TerrainCfg:
    """Field-for-field mirror of the reference's `TerrainCfg` (cfg/terrain/terrain_cfg.py:1-22)."""
    mesh_type = "plane"            # none, plane, heightfield or trimesh
    horizontal_scale = 0.1         # [m]
    vertical_scale = 0.005         # [m]
    border_size = 25               # [m]
    curriculum = False
    static_friction = 1.0
    dynamic_friction = 1.0
    restitution = 0.0
    selected = False
    terrain_kwargs = None
    max_init_terrain_level = 5
    terrain_length = 8.0
    terrain_width = 8.0
    num_rows = 10                  # terrain rows (levels)
    num_cols = 20                  # terrain columns (types)
    # [smooth slope, rough slope, stairs up, stairs down, discrete]
    terrain_proportions = [0.1, 0.1, 0.35, 0.25, 0.2]
    slope_treshold = 0.75

    def __init__(self, **overrides):
        for k, v in overrides.items():
            if not hasattr(type(self), k):
                raise AttributeError("TerrainCfg has no field %r" % k)
            setattr(self, k, v)


class Tile:
    """One sub-terrain: `width` x `length` int16 samples."""

    def __init__(self, width: int, length: int, vertical_scale: float, horizontal_scale: float):
        self.width, self.length = int(width), int(length)
        self.vertical_scale, self.horizontal_scale = vertical_scale, horizontal_scale
        self.height_field_raw = np.zeros((self.width, self.length), dtype=np.int16)


# ---------------------------------------------------------------------------------------------- tile shapes
def pyramid_slope(t: Tile, slope: float, platform_size: float = 1.0) -> None:
    """Four-sided ramp towards a flat square platform (terrain_utils.py:74-106)."""
    cx, cy = int(t.width / 2), int(t.length / 2)
    fx = ((cx - np.abs(cx - np.arange(t.width))) / cx).reshape(t.width, 1)
    fy = ((cy - np.abs(cy - np.arange(t.length))) / cy).reshape(1, t.length)
    peak = int(slope * (t.horizontal_scale / t.vertical_scale) * (t.width / 2))
    t.height_field_raw += (peak * fx * fy).astype(t.height_field_raw.dtype)
    half = int(platform_size / t.horizontal_scale / 2)
    x1, y1 = t.width // 2 - half, t.length // 2 - half
    edge = t.height_field_raw[x1, y1]
    t.height_field_raw = np.clip(t.height_field_raw, min(edge, 0), max(edge, 0))


def _bilinear_resample(coarse: np.ndarray, n_out_x: int, n_out_y: int) -> np.ndarray:
    """Values of the piecewise-bilinear interpolant of `coarse` (nodes spread evenly over the same extent) at
    n_out_x x n_out_y evenly spread points -- what `interp2d(kind='linear')` returned on a regular grid."""
    nx, ny = coarse.shape
    gx = np.linspace(0.0, nx - 1.0, n_out_x)
    gy = np.linspace(0.0, ny - 1.0, n_out_y)
    ix = np.clip(np.floor(gx).astype(int), 0, nx - 2)
    iy = np.clip(np.floor(gy).astype(int), 0, ny - 2)
    tx = (gx - ix).reshape(-1, 1)
    ty = (gy - iy).reshape(1, -1)
    c = coarse.astype(np.float64)
    c00 = c[np.ix_(ix, iy)]
    c10 = c[np.ix_(ix + 1, iy)]
    c01 = c[np.ix_(ix, iy + 1)]
    c11 = c[np.ix_(ix + 1, iy + 1)]
    return (1 - tx) * (1 - ty) * c00 + tx * (1 - ty) * c10 + (1 - tx) * ty * c01 + tx * ty * c11


def random_uniform(t: Tile, rs: np.random.RandomState, min_height: float, max_height: float, step: float = 1.0,
                   downsampled_scale: float | None = None) -> None:
    """Uniform noise drawn on a coarse grid and interpolated up (terrain_utils.py:17-51)."""
    if downsampled_scale is None:
        downsampled_scale = t.horizontal_scale
    lo, hi, st = int(min_height / t.vertical_scale), int(max_height / t.vertical_scale), int(step / t.vertical_scale)
    levels = np.arange(lo, hi + st, st)
    coarse = rs.choice(levels, (int(t.width * t.horizontal_scale / downsampled_scale),
                                int(t.length * t.horizontal_scale / downsampled_scale)))
    t.height_field_raw += np.rint(_bilinear_resample(coarse, t.width, t.length)).astype(np.int16)


def pyramid_stairs(t: Tile, step_width: float, step_height: float, platform_size: float = 1.0) -> None:
    """Concentric square steps up (or down, negative height) to a platform (terrain_utils.py:195-224)."""
    sw, sh = int(step_width / t.horizontal_scale), int(step_height / t.vertical_scale)
    plat = int(platform_size / t.horizontal_scale)
    x0, x1, y0, y1, h = 0, t.width, 0, t.length, 0
    while (x1 - x0) > plat and (y1 - y0) > plat:
        x0 += sw; x1 -= sw; y0 += sw; y1 -= sw
        h += sh
        t.height_field_raw[x0:x1, y0:y1] = h


def discrete_obstacles(t: Tile, rs: np.random.RandomState, max_height: float, min_size: float, max_size: float,
                       num_rects: int, platform_size: float = 1.0) -> None:
    """Random rectangular blocks and pits around a flat platform (terrain_utils.py:109-146)."""
    mh = int(max_height / t.vertical_scale)
    lo, hi = int(min_size / t.horizontal_scale), int(max_size / t.horizontal_scale)
    plat = int(platform_size / t.horizontal_scale)
    ni, nj = t.height_field_raw.shape
    heights = [-mh, -mh // 2, mh // 2, mh]
    sizes = range(lo, hi, 4)
    for _ in range(num_rects):
        w = rs.choice(sizes)
        ln = rs.choice(sizes)
        i0 = rs.choice(range(0, ni - w, 4))
        j0 = rs.choice(range(0, nj - ln, 4))
        t.height_field_raw[i0:i0 + w, j0:j0 + ln] = rs.choice(heights)
    x1, x2 = (t.width - plat) // 2, (t.width + plat) // 2
    y1, y2 = (t.length - plat) // 2, (t.length + plat) // 2
    t.height_field_raw[x1:x2, y1:y2] = 0


def stepping_stones(t: Tile, rs: np.random.RandomState, stone_size: float, stone_distance: float, max_height: float,
                    platform_size: float = 1.0, depth: float = -10.0) -> None:
    """Square stones over a deep pit (terrain_utils.py:227-283)."""
    ss, sd = int(stone_size / t.horizontal_scale), int(stone_distance / t.horizontal_scale)
    mh, plat = int(max_height / t.vertical_scale), int(platform_size / t.horizontal_scale)
    heights = np.arange(-mh - 1, mh, step=1)
    f = t.height_field_raw
    f[:, :] = int(depth / t.vertical_scale)
    if t.length >= t.width:
        y = 0
        while y < t.length:
            y_end = min(t.length, y + ss)
            x = rs.randint(0, ss)
            f[0:max(0, x - sd), y:y_end] = rs.choice(heights)
            while x < t.width:
                f[x:min(t.width, x + ss), y:y_end] = rs.choice(heights)
                x += ss + sd
            y += ss + sd
    else:
        x = 0
        while x < t.width:
            x_end = min(t.width, x + ss)
            y = rs.randint(0, ss)
            f[x:x_end, 0:max(0, y - sd)] = rs.choice(heights)
            while y < t.length:
                f[x:x_end, y:min(t.length, y + ss)] = rs.choice(heights)
                y += ss + sd
            x += ss + sd
    x1, x2 = (t.width - plat) // 2, (t.width + plat) // 2
    y1, y2 = (t.length - plat) // 2, (t.length + plat) // 2
    f[x1:x2, y1:y2] = 0


def gap(t: Tile, gap_size: float, platform_size: float = 1.0) -> None:
    """A moat around the platform (utils/terrain.py:166-178)."""
    g, plat = int(gap_size / t.horizontal_scale), int(platform_size / t.horizontal_scale)
    cx, cy = t.length // 2, t.width // 2
    x1 = (t.length - plat) // 2
    y1 = (t.width - plat) // 2
    x2, y2 = x1 + g, y1 + g
    t.height_field_raw[cx - x2:cx + x2, cy - y2:cy + y2] = -1000
    t.height_field_raw[cx - x1:cx + x1, cy - y1:cy + y1] = 0


def pit(t: Tile, depth: float, platform_size: float = 1.0) -> None:
    """A sunken platform (utils/terrain.py:180-188)."""
    d, half = int(depth / t.vertical_scale), int(platform_size / t.horizontal_scale / 2)
    x1, x2 = t.length // 2 - half, t.length // 2 + half
    y1, y2 = t.width // 2 - half, t.width // 2 + half
    t.height_field_raw[x1:x2, y1:y2] = -d


def heightfield_to_trimesh(samples: np.ndarray, horizontal_scale: float, vertical_scale: float,
                           slope_threshold: float | None = None):
    """Vertices [rows*cols, 3] (float32, metres) and triangles [2(rows-1)(cols-1), 3] (uint32) of the grid, two triangles
    per cell split along the (i,j)-(i+1,j+1) diagonal; with a slope threshold, the lower vertex of every edge or cell
    diagonal steeper than it is pushed under the upper one, which turns steep ramps into vertical faces
    (terrain_utils.py:286-330).  Only exposed for viewers and exporters: the kernels collide with the samples."""
    hf = samples
    rows, cols = hf.shape
    yy, xx = np.meshgrid(np.linspace(0, (cols - 1) * horizontal_scale, cols), np.linspace(0, (rows - 1) * horizontal_scale, rows))
    if slope_threshold is not None:
        thr = slope_threshold * horizontal_scale / vertical_scale
        mx, my, mc = np.zeros((rows, cols)), np.zeros((rows, cols)), np.zeros((rows, cols))
        dx = hf[1:, :] - hf[:-1, :]
        dy = hf[:, 1:] - hf[:, :-1]
        dd = hf[1:, 1:] - hf[:-1, :-1]
        mx[:-1, :] += dx > thr
        mx[1:, :] -= -dx > thr
        my[:, :-1] += dy > thr
        my[:, 1:] -= -dy > thr
        mc[:-1, :-1] += dd > thr
        mc[1:, 1:] -= -dd > thr
        xx += (mx + mc * (mx == 0)) * horizontal_scale
        yy += (my + mc * (my == 0)) * horizontal_scale
    vertices = np.zeros((rows * cols, 3), dtype=np.float32)
    vertices[:, 0], vertices[:, 1], vertices[:, 2] = xx.ravel(), yy.ravel(), hf.ravel() * vertical_scale
    i0 = (np.arange(rows - 1)[:, None] * cols + np.arange(cols - 1)[None, :]).ravel()
    tri = np.empty((2 * (rows - 1) * (cols - 1), 3), dtype=np.uint32)
    tri[0::2] = np.stack([i0, i0 + cols + 1, i0 + 1], 1)
    tri[1::2] = np.stack([i0, i0 + cols, i0 + cols + 1], 1)
    return vertices, tri


# ---------------------------------------------------------------------------------------------- the map
class Terrain:
    """The tiled map (reference: utils/terrain.py:40-165).  Attributes follow the reference: `heightsamples`
    (= `height_field_raw`, int16 [tot_rows, tot_cols]), `env_origins` [num_rows, num_cols, 3], `tot_rows`, `tot_cols`,
    `border`, `env_length`, `env_width`."""

    def __init__(self, cfg: TerrainCfg, num_robots: int, seed: int | None = None):
        self.cfg, self.num_robots, self.type = cfg, num_robots, cfg.mesh_type
        if self.type in ("none", "plane", None):
            return
        if cfg.selected:
            raise NotImplementedError("TerrainCfg.selected: pass terrain_proportions that select the type instead")
        self.rs = np.random.RandomState(seed)
        self.env_length, self.env_width = cfg.terrain_length, cfg.terrain_width
        self.proportions = [float(np.sum(cfg.terrain_proportions[:i + 1])) for i in range(len(cfg.terrain_proportions))]
        self.env_origins = np.zeros((cfg.num_rows, cfg.num_cols, 3))
        self.width_per_env_pixels = int(self.env_width / cfg.horizontal_scale)
        self.length_per_env_pixels = int(self.env_length / cfg.horizontal_scale)
        self.border = int(cfg.border_size / cfg.horizontal_scale)
        self.tot_cols = int(cfg.num_cols * self.width_per_env_pixels) + 2 * self.border
        self.tot_rows = int(cfg.num_rows * self.length_per_env_pixels) + 2 * self.border
        self.height_field_raw = np.zeros((self.tot_rows, self.tot_cols), dtype=np.int16)
        if cfg.curriculum:
            for j in range(cfg.num_cols):
                for i in range(cfg.num_rows):
                    self._place(self.make_tile(j / cfg.num_cols + 0.001, i / cfg.num_rows), i, j)
        else:
            for k in range(cfg.num_rows * cfg.num_cols):
                i, j = np.unravel_index(k, (cfg.num_rows, cfg.num_cols))
                choice = self.rs.uniform(0, 1)
                difficulty = self.rs.choice([0.5, 0.75, 0.9])
                self._place(self.make_tile(choice, difficulty), i, j)
        self.heightsamples = self.height_field_raw
        if self.type == "trimesh":
            self.vertices, self.triangles = heightfield_to_trimesh(self.height_field_raw, cfg.horizontal_scale,
                                                                   cfg.vertical_scale, cfg.slope_treshold)

    def make_tile(self, choice: float, difficulty: float) -> Tile:
        """Tile type from `choice` against the cumulative proportions, size of its features from `difficulty`
        (utils/terrain.py:106-145)."""
        c = self.cfg
        t = Tile(self.width_per_env_pixels, self.width_per_env_pixels, c.vertical_scale, c.horizontal_scale)
        p = self.proportions + [np.inf] * (7 - len(self.proportions))
        slope = difficulty * 0.4
        if choice < p[0]:
            pyramid_slope(t, -slope if choice < p[0] / 2 else slope, platform_size=3.0)
        elif choice < p[1]:
            pyramid_slope(t, slope, platform_size=3.0)
            random_uniform(t, self.rs, -0.05, 0.05, step=0.005, downsampled_scale=0.2)
        elif choice < p[3]:
            step_height = 0.02 + 0.2 * difficulty
            if choice < p[2]:
                step_height = -(0.01 + 0.10 * difficulty)
            pyramid_stairs(t, step_width=0.5, step_height=step_height, platform_size=3.0)
        elif choice < p[4]:
            discrete_obstacles(t, self.rs, 0.01 + difficulty * 0.10, 1.0, 2.0, 20, platform_size=3.0)
        elif choice < p[5]:
            stepping_stones(t, self.rs, stone_size=1.5 * (1.05 - difficulty), stone_distance=0.05 if difficulty == 0 else 0.1,
                            max_height=0.0, platform_size=4.0)
        elif choice < p[6]:
            gap(t, gap_size=1.0 * difficulty, platform_size=3.0)
        else:
            pit(t, depth=1.0 * difficulty, platform_size=4.0)
        return t

    def _place(self, t: Tile, i: int, j: int) -> None:
        x0 = self.border + i * self.length_per_env_pixels
        y0 = self.border + j * self.width_per_env_pixels
        self.height_field_raw[x0:x0 + self.length_per_env_pixels, y0:y0 + self.width_per_env_pixels] = t.height_field_raw
        hs = t.horizontal_scale
        x1, x2 = int((self.env_length / 2.0 - 1) / hs), int((self.env_length / 2.0 + 1) / hs)
        y1, y2 = int((self.env_width / 2.0 - 1) / hs), int((self.env_width / 2.0 + 1) / hs)
        z = np.max(t.height_field_raw[x1:x2, y1:y2]) * t.vertical_scale
        self.env_origins[i, j] = [(i + 0.5) * self.env_length, (j + 0.5) * self.env_width, z]

    # ---- what the physics needs ----
    def height_at(self, x: np.ndarray, y: np.ndarray) -> np.ndarray:
        """Bilinear terrain height [m] at world points (the sampling rule of the kernels, in float64)."""
        c = self.cfg
        u = (np.asarray(x, dtype=np.float64) + c.border_size) / c.horizontal_scale
        v = (np.asarray(y, dtype=np.float64) + c.border_size) / c.horizontal_scale
        u = np.clip(u, 0.0, self.tot_rows - 1.000001)
        v = np.clip(v, 0.0, self.tot_cols - 1.000001)
        i, j = np.floor(u).astype(int), np.floor(v).astype(int)
        a, b = u - i, v - j
        h = self.height_field_raw.astype(np.float64)
        return c.vertical_scale * ((1 - a) * (1 - b) * h[i, j] + a * (1 - b) * h[i + 1, j] +
                                   (1 - a) * b * h[i, j + 1] + a * b * h[i + 1, j + 1])

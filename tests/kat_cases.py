"""Hand-derived known answers for octree_ray (shader.wgsl:191-248) beyond the one-level tree of SURVEY.md 8c.

The reference holds no test vectors and cannot be run here, so these are the independent pin of the oracle: every expected
record below was worked out on paper from the WGSL alone -- the arithmetic is written out next to each case, all of it exact in
f32 (dyadic rationals) on the axis that decides -- and the SAME literals are asserted on the CPU oracle
(tests/test_oracle_kat.py) and on every GPU kernel variant through svo_trace_rays (tests/test_parity_gpu.py).

Conventions (shader.wgsl): child index = 4*(x > cx) + 2*(y > cy) + (z > cz) (`>=` with misc_bool, :138-150); a zero direction
component is replaced by 1e-6 (:193-194); a ray that starts outside enters at pos + dir * ray_box_dist (:197-205, the box test
uses the unbiased direction :66-80); per step t_max = (centre - pos + sign(dir) * size / 2) / dir, the step takes the smallest
t_max, ALL axes that attain it (`<=`, :231) go into the normal, voxel_pos = pos + dir * t - normal * 2e-6 (:227-235); leaving
the cube returns value 0x20202000 WITHOUT counting the step (:236-239); the count is checked after the increment: steps > 100
returns value 0xFF000000, depth 100, hit = true (:241-244).  Record t = entry distance + t of the last step.
normal_bits: 2 bits per axis (x: 0..1, y: 2..3, z: 4..5), 1 -> +1, 2 -> -1.
"""
import numpy as np

V = 134217728  # VOXEL_OFFSET (octree.rs:5)
EMPTY = V << 4
MISS_LEFT = 0x20202000
STEP_LIMIT = 0xFF000000
F_PAUSE_ADAPTIVE, F_MISC_BOOL = 1, 16


def solid(rgb):
    return (V + rgb) << 4


def column_tree(solid_cells):
    """A tree whose leaves along the line x = y = 0.99 are the 128 cells of level 7 (size 1/64), cell k covering
    z in [-1 + k/64, -1 + (k+1)/64): on that line every level chooses x-bit 1 and y-bit 1, so the child index is 6 (lower z half)
    or 7 (upper).  Node (l, j) -- level l = 0 (root) .. 6, j = 0 .. 2^l - 1 counted along z -- owns the child group at word
    8 * (2^l - 1 + j); its children 0..5 are empty leaves, children 6 and 7 are the nodes (l+1, 2j) and (l+1, 2j+1), which for
    l = 6 are the leaves of cells 2j and 2j+1.  So cell k is word 8 * (63 + k // 2) + 6 + k % 2; 127 groups = 1016 words."""
    words = np.full(8 * 127, EMPTY, dtype=np.uint32)
    for l in range(7):
        for j in range(1 << l):
            g = 8 * ((1 << l) - 1 + j)
            for half in (0, 1):
                if l < 6:
                    words[g + 6 + half] = (8 * ((1 << (l + 1)) - 1 + 2 * j + half)) << 4
                else:
                    k = 2 * j + half
                    words[g + 6 + half] = solid(0x0000FF + k) if k in solid_cells else EMPTY
    return words


def cell_word(k):
    return 8 * (63 + k // 2) + 6 + k % 2


def b_tree():
    """one level, child 7 solid"""
    b = np.full(8, EMPTY, dtype=np.uint32)
    b[7] = solid(0xFF0000)
    return b


def rec(value, t, steps, depth, hit, normal_bits):
    return dict(value=value, t=np.float32(t), steps=steps, depth=depth, hit=hit, normal_bits=normal_bits)


def cases():
    """[(name, words, flags, rays (n, 6) float32, [expected record per ray])]"""
    out = []
    # ---- A: two levels.  Root child 6 (x>0, y>0, z<=0) is interior -> group at word 8; root child 7 is solid.
    # Ray 1: pos (0.5, 0.5, -3), dir (0, 0, 1).  Box: z slab [(-1+3)/1, (1+3)/1] = [2, 4], x/y slabs are [-inf, inf] -> dist 2;
    #   entry (0.5 + 2e-6, 0.5 + 2e-6, -1) (x, y moved by the 1e-6 bias: strictly above the planes x = 0.5, y = 0.5).
    #   Walk 1: root: z = -1 > 0 no -> child 6 -> group 8, centre (0.5, 0.5, -0.5): z > -0.5 no -> child 6 = word 14, depth 2,
    #     centre z -0.75, size 0.5, empty.  t_z = (-0.75 + 1 + 0.25) / 1 = 0.5 (t_x, t_y ~ 5e5).  voxel_pos z = -1 + 0.5 + 2e-6.
    #   Walk 2: root child 6 again; z = -0.499998 > -0.5 yes -> child 7 = word 15, centre z -0.25, empty.
    #     t_z = (-0.25 + 1 + 0.25) / 1 = 1.  voxel_pos z = -1 + 1 + 2e-6 = 2e-6.
    #   Walk 3: root: z > 0 yes -> child 7 = word 7, solid, depth 1.  steps 2, t = 2 + 1 = 3, normal (0, 0, -1).
    # Ray 2: pos (0.5, 0.5, 3), dir (0, 0, -1): dist 2, entry z = 3 - 2 = 1 exactly; root: z > 0 -> word 7, solid at once:
    #   steps 0, t = 2 + 0, normal = trunc(entry * 1.000001) = (0, 0, 1).
    a = np.full(16, EMPTY, dtype=np.uint32)
    a[6] = 8 << 4
    a[7] = solid(0xFF0000)
    out.append(("two levels, axis ray", a, F_PAUSE_ADAPTIVE,
                np.array([[0.5, 0.5, -3, 0, 0, 1], [0.5, 0.5, 3, 0, 0, -1]], dtype=np.float32),
                [rec(7, 3.0, 2, 1, 1, 2 << 4), rec(7, 2.0, 0, 1, 1, 1 << 4)]))
    # ---- A': the same with word 15 solid: ray 1 ends in walk 2: value 15, depth 2, steps 1, t = 2 + 0.5.
    a2 = a.copy()
    a2[15] = solid(0x00FF00)
    out.append(("two levels, hit at depth 2", a2, F_PAUSE_ADAPTIVE, np.array([[0.5, 0.5, -3, 0, 0, 1]], dtype=np.float32),
                [rec(15, 2.5, 1, 2, 1, 2 << 4)]))
    # ---- A'': going down (-z) through the column with root child 7 EMPTY: pos (0.5, 0.5, 3), dir (0, 0, -1), dist 2, entry z = 1.
    #   Walk 1: child 7 (depth 1, centre z 0.5, size 1): t_z = (0.5 - 1 + (-1) * 0.5) / (-1) = 1; the normal is +z, so
    #     voxel_pos z = 1 - 1 - 2e-6 = -2e-6.  Walk 2: child 6 -> group 8, z > -0.5 -> word 15 (centre z -0.25, size 0.5):
    #     t_z = (-0.25 - 1 - 0.25) / (-1) = 1.5; voxel_pos z = -0.5 - 2e-6.  Walk 3: z > -0.5 no -> word 14 (centre z -0.75):
    #     t_z = (-0.75 - 1 - 0.25) / (-1) = 2; voxel_pos z = -1 - 2e-6: left the cube after two counted steps: t = 2 + 2, depth 2.
    a3 = a.copy()
    a3[7] = EMPTY
    out.append(("two levels, downwards, leaves through the bottom", a3, F_PAUSE_ADAPTIVE, np.array([[0.5, 0.5, 3, 0, 0, -1]], dtype=np.float32),
                [rec(MISS_LEFT, 4.0, 2, 2, 0, 0)]))
    # ---- F: the cube's faces and the box test (shader.wgsl:66-80, 177-180, 197-205), one level, child 7 solid.
    #   Ray 1: pos (1.5, 0.25, 0.25), dir (-1, 0, 0): x slab [(1 - 1.5) / -1, (-1 - 1.5) / -1] = [0.5, 2.5], y / z slabs [-inf, inf] ->
    #     dist 0.5; entry (1.5 - 0.5, 0.25 + 1e-6 * 0.5, same) = (1, 0.2500005, 0.2500005): every component > 0 -> child 7, solid at
    #     once: steps 0, t = 0.5, normal = trunc((1, 0.25, 0.25) * 1.000001) = (1, 0, 0).
    #   Ray 2: the same ray started ON the face: pos (1, 0.25, 0.25).  in_bounds wants x < 1, so the box test runs: x slab
    #     [(1 - 1) / -1, (-1 - 1) / -1] = [-0, 2] -> v7 = -0, v8 = 2: ray_box_dist returns -0.0, which octree_ray reads as "0 = no
    #     intersection" (dist == 0.0): the ray never enters -- the all-zero record.  (A quirk of the reference, kept.)
    #   Ray 3: parallel to z beside the cube: pos (2, 0.5, -3), dir (0, 0, 1): x slab [(-1 - 2) / 0, (1 - 2) / 0] = [-inf, -inf],
    #     v8 = -inf < 0 -> 0: never enters.
    out.append(("faces and the box test", b_tree(), F_PAUSE_ADAPTIVE,
                np.array([[1.5, 0.25, 0.25, -1, 0, 0], [1.0, 0.25, 0.25, -1, 0, 0], [2, 0.5, -3, 0, 0, 1]], dtype=np.float32),
                [rec(7, 0.5, 0, 1, 1, 1), rec(0, 0.0, 0, 0, 0, 0), rec(0, 0.0, 0, 0, 0, 0)]))
    # ---- B: three-axis tie.  One level, child 7 solid.  pos (-3, -3, -3), dir (1, 1, 1) (not normalised: octree_ray takes it as
    #   it is).  Every slab is [2, 4] -> dist 2, entry (-1, -1, -1): child 0, centre -0.5, size 1, empty.
    #   t_x = t_y = t_z = (-0.5 + 1 + 0.5) / 1 = 1: all three attain the minimum -> normal (-1, -1, -1);
    #   voxel_pos = -1 + 1 + 2e-6 on every axis -> child 7, solid: steps 1, depth 1, t = 2 + 1, normal bits 2 | 2<<2 | 2<<4.
    # ---- C: two-axis tie.  pos (-3, -3, 0.5), dir (1, 1, 0): z slab [-inf, inf], dist 2, entry (-1, -1, 0.5 + 2e-6): child 1
    #   (z > 0), centre (-0.5, -0.5, 0.5), empty.  t_x = t_y = 1, t_z = (0.5 - 0.500002 + 0.5) / 1e-6 ~ 5e5 -> normal (-1, -1, 0);
    #   voxel_pos (2e-6, 2e-6, 0.500003) -> child 7: steps 1, t = 3, normal bits 2 | 2<<2.
    b = np.full(8, EMPTY, dtype=np.uint32)
    b[7] = solid(0xFF0000)
    out.append(("three- and two-axis ties", b, F_PAUSE_ADAPTIVE,
                np.array([[-3, -3, -3, 1, 1, 1], [-3, -3, 0.5, 1, 1, 0]], dtype=np.float32),
                [rec(7, 3.0, 1, 1, 1, 2 | (2 << 2) | (2 << 4)), rec(7, 3.0, 1, 1, 1, 2 | (2 << 2))]))
    # ---- B': the diagonal through an EMPTY tree: second step from child 7 (centre 0.5): t = (0.5 + 1 + 0.5) / 1 = 2,
    #   voxel_pos = -1 + 2 + 2e-6 > 1: left the cube: value 0x20202000, the step is not counted (steps 1), depth 1, t = 2 + 2.
    out.append(("diagonal through an empty cube", np.full(8, EMPTY, dtype=np.uint32), F_PAUSE_ADAPTIVE,
                np.array([[-3, -3, -3, 1, 1, 1]], dtype=np.float32), [rec(MISS_LEFT, 4.0, 1, 1, 0, 0)]))
    # ---- D: a start point ON a centre plane.  One level, child 2 = (x<=0, y>0, z<=0) and child 6 = (x>0, y>0, z<=0) solid.
    #   pos (0, 0.25, -0.75) is inside the cube (dist 0, pos unchanged), dir (0, 0, 1).  x = 0 against the plane x = 0:
    #   `>` (default) says no -> child 2; `>=` (misc_bool) says yes -> child 6.  Solid at once: steps 0, t 0,
    #   normal = trunc(pos * 1.000001) = (0, 0, 0).
    d = np.full(8, EMPTY, dtype=np.uint32)
    d[2] = solid(0x0000AA)
    d[6] = solid(0x00BB00)
    ray_d = np.array([[0.0, 0.25, -0.75, 0, 0, 1]], dtype=np.float32)
    out.append(("on a centre plane, >", d, F_PAUSE_ADAPTIVE, ray_d, [rec(2, 0.0, 0, 1, 1, 0)]))
    out.append(("on a centre plane, >=", d, F_PAUSE_ADAPTIVE | F_MISC_BOOL, ray_d, [rec(6, 0.0, 0, 1, 1, 0)]))
    # ---- E: the step limit, on column_tree.  pos (0.99, 0.99, -3), dir (0, 0, 1): dist 2, entry z = -1, x = y = 0.99 + 2e-6.
    #   From cell k: t_z = (centre_k + 1 + 1/128) / 1 = ((2k+1)/128 + 1/128) = (k+1)/64 exactly (t_x, t_y ~ 1e4);
    #   voxel_pos z = -1 + (k+1)/64 + 2e-6 lies in cell k+1.  After j steps the ray is in cell j.
    #   Solid cell 100: reached with steps = 100, and 100 > 100 is false -> an ordinary hit: value = its word, depth 7,
    #     t = 2 + 100/64 = 3.5625.
    #   Solid cell 101: the 101st step makes steps = 101 > 100 -> value 0xFF000000, steps 101, depth 100, hit,
    #     t = 2 + 101/64 = 3.578125, normal of that step (0, 0, -1).
    up = np.array([[0.99, 0.99, -3, 0, 0, 1]], dtype=np.float32)
    out.append(("100 steps, then a hit", column_tree({100}), F_PAUSE_ADAPTIVE, up, [rec(cell_word(100), 3.5625, 100, 7, 1, 2 << 4)]))
    out.append(("101 steps: the limit", column_tree({101}), F_PAUSE_ADAPTIVE, up, [rec(STEP_LIMIT, 3.578125, 101, 100, 1, 2 << 4)]))
    #   Leaving the cube on the empty column, from a start INSIDE it (dist 0): start at the centre of cell c, z0 = -1 + (2c+1)/128.
    #   From cell k: t_z = ((2k+1)/128 - (2c+1)/128 + 1/128) = (2k - 2c + 1)/128.
    #   c = 27: cells 27..127 are 101 cells; the 101st step (k = 127: t = 201/128, voxel_pos z = 1 + 2e-6) leaves the cube with
    #     100 steps counted: value 0x20202000, steps 100, depth 7, no hit, t = 1.5703125.
    #   c = 26: 102 cells; the 101st step (k = 126: t = 201/128) stays inside, steps = 101 > 100: the limit record.
    z27, z26 = -1 + 55 / 128, -1 + 53 / 128
    out.append(("leaves the cube with 100 steps counted / needs 102", column_tree(set()), F_PAUSE_ADAPTIVE,
                np.array([[0.99, 0.99, z27, 0, 0, 1], [0.99, 0.99, z26, 0, 0, 1]], dtype=np.float32),
                [rec(MISS_LEFT, 1.5703125, 100, 7, 0, 0), rec(STEP_LIMIT, 1.5703125, 101, 100, 1, 2 << 4)]))
    return out


def check(hits, expected, what):
    """hits: records (oracle or GPU) in HIT_DTYPE; expected: list of rec()."""
    for i, e in enumerate(expected):
        h = hits[i]
        got = dict(value=int(h["value"]), t=np.float32(h["t"]), steps=int(h["info"] & 0xFF), depth=int((h["info"] >> 8) & 0xFF),
                   hit=int((h["info"] >> 16) & 1), normal_bits=int(h["normal_bits"]))
        assert got == e, f"{what}, ray {i}: got {got}, derived by hand {e}"


def one_ray_camera_inverse(pos, direction):
    """camera_inverse (column-major) of a 1 x 1 frame whose only ray is exactly (pos, direction): the pixel centre is clip (0, 0)
    (shader.wgsl:253-259: (0.5 / 1 * 2 - 1) * (1, -1)), so origin = (Cinv * (0, 0, 0, 1)).xyz / w = column 3 and
    direction = normalize((Cinv * (0, 0, 1, 1)).xyz / w - origin) = normalize(column 2) (shader.wgsl:54-59).  `direction` must be of
    length exactly 1 and pos + direction exactly representable for the ray to be exact."""
    m = np.zeros(16, dtype=np.float32)
    m[0] = 1.0                                  # column 0 = (1, 0, 0, 0)
    m[5] = 1.0                                  # column 1 = (0, 1, 0, 0)
    m[8:11] = direction                         # column 2 = (d, 0)
    m[12:15] = pos                              # column 3 = (p, 1)
    m[15] = 1.0
    return m


def counter_case():
    """Hit counters (shader.wgsl:157-161) derived by hand.  find_voxel bumps EVERY word it reads -- interior words and the leaf
    alike, before the leaf test -- once per call, while `primary && counter < 15 && !pause_adaptive`.  Tree A of cases() (two
    levels), the one ray (0.5, 0.5, -3) -> (0, 0, 1), counters live, no shadow ray: walk 1 reads words 6 and 14, walk 2 words 6 and
    15, walk 3 word 7 (the hit).  After n frames: word 6 carries min(15, 2n), words 14, 15 and 7 min(15, n); nothing else changes.
    Returns (words, camera_inverse, record of the ray, function n -> expected words)."""
    name, words, flags, rays, expected = cases()[0]
    cinv = one_ray_camera_inverse(rays[0, :3], rays[0, 3:])

    def after(n):
        w = words.copy()
        w[6] += min(15, 2 * n)
        for i in (14, 15, 7):
            w[i] += min(15, n)
        return w
    return words, cinv, expected[0], after


def shading_cases():
    """fs_main (shader.wgsl:261-304) by hand, on 1 x 1 frames of tree A (cases()[0]).
    Hit pixel: the ray (0.5, 0.5, -3) -> (0, 0, 1) ends on word 7, colour (255, 0, 0), normal (0, 0, -1).  sun_dir (-1.7, -1, 0.8),
    |sun| = sqrt(2.89 + 1 + 0.64) = 2.12838; diffuse = max(dot(n, -sun / |sun|), 0) = 0.8 / 2.12838 = 0.37587; the shadow ray (from the hit
    point, 5e-7 below z = 0, towards (0.799, 0.470, -0.376)) crosses only empty cells and leaves through the face x = 1 at t = 0.63: lit.
    colour = (0.3 + 0.37587) * (1, 0, 0) = 0.67587; 0.67587 ^ 2.2 = 0.42238 -> 108.2 -> 108; alpha 0.5 -> 128.
    Miss pixel: the same origin looking away (0, 0, -1) never enters the cube: 0.2 grey, 0.2 ^ 2.2 = 0.02899 -> 7.9 -> 7.
    Returns [(name, words, flags, camera_inverse, sun_dir, expected RGBA8)]; pow() may move a channel by one code value."""
    name, words, flags, rays, expected = cases()[0]
    sun = (-1.7, -1.0, 0.8, 0.0)
    F_SHADOWS = 8
    hit = one_ray_camera_inverse(rays[0, :3], rays[0, 3:])
    away = one_ray_camera_inverse(rays[0, :3], np.array([0, 0, -1], dtype=np.float32))
    return [("lit hit, no shadow ray", words, F_PAUSE_ADAPTIVE, hit, sun, (108, 0, 0, 128)),
            ("lit hit, shadow ray finds nothing", words, F_PAUSE_ADAPTIVE | F_SHADOWS, hit, sun, (108, 0, 0, 128)),
            ("miss", words, F_PAUSE_ADAPTIVE | F_SHADOWS, away, sun, (7, 7, 7, 128))]

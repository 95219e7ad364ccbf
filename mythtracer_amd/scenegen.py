"""Deterministic synthetic scenes (Wavefront .obj/.mtl text).

The model the reference is demonstrated on ("Living Room USSU Design",
/root/reference/README.md:4-5, VerStarting/main_local.cc:35) is not in the
reference repository and cannot be downloaded, so BASELINE.json's "living-room
.obj (~100k tris)" is replaced by a procedural room of about the same size and
make-up: large tessellated walls, many finely tessellated round objects and
box furniture; diffuse, reflective (Refl), mirror and glass (Tr/Tf) materials.

Only + - * / sqrt and a splitmix64 integer generator are used, and every number
is printed with %.6f, so the text is bit-reproducible on any machine.  Every
`f` line ends with a space: the reference's face parser drops the last index of
a line that has no trailing whitespace (VerStarting/objreader.cc:111-115).
"""
from __future__ import annotations

import hashlib
import math
import os

MASK = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed: int):
        self.s = seed & MASK

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
        return z ^ (z >> 31)

    def unit(self) -> float:
        """Uniform in [0,1) with 20 bits, exactly representable."""
        return (self.next() >> 44) / 1048576.0

    def rng(self, lo: float, hi: float) -> float:
        return lo + (hi - lo) * self.unit()


def _icosphere(level: int):
    """Unit icosphere: (vertices, faces) by midpoint subdivision (sqrt only)."""
    t = (1.0 + math.sqrt(5.0)) / 2.0
    raw = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0),
           (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
           (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    verts = []
    for x, y, z in raw:
        l = math.sqrt(x * x + y * y + z * z)
        verts.append((x / l, y / l, z / l))
    faces = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11),
             (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
             (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9),
             (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    for _ in range(level):
        cache = {}

        def mid(a, b):
            key = (a, b) if a < b else (b, a)
            if key in cache:
                return cache[key]
            x = (verts[a][0] + verts[b][0]) / 2.0
            y = (verts[a][1] + verts[b][1]) / 2.0
            z = (verts[a][2] + verts[b][2]) / 2.0
            l = math.sqrt(x * x + y * y + z * z)
            verts.append((x / l, y / l, z / l))
            cache[key] = len(verts) - 1
            return cache[key]

        nf = []
        for a, b, c in faces:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        faces = nf
    return verts, faces


_ICO_CACHE: dict = {}


def icosphere(level: int):
    if level not in _ICO_CACHE:
        _ICO_CACHE[level] = _icosphere(level)
    return _ICO_CACHE[level]


class ObjWriter:
    """Accumulates OBJ text; indices are 1-based as the format wants."""

    def __init__(self, mtllib: str | None):
        self.lines = []
        if mtllib:
            self.lines.append("mtllib %s" % mtllib)
        self.nv = 0
        self.nn = 0
        self.nt = 0
        self.ntris = 0
        self.use_mtl = mtllib is not None

    def usemtl(self, name: str):
        if self.use_mtl:
            self.lines.append("usemtl %s" % name)

    def v(self, x, y, z) -> int:
        self.lines.append("v %.6f %.6f %.6f" % (x, y, z))
        self.nv += 1
        return self.nv

    def vn(self, x, y, z) -> int:
        self.lines.append("vn %.6f %.6f %.6f" % (x, y, z))
        self.nn += 1
        return self.nn

    def vt(self, u, v) -> int:
        self.lines.append("vt %.6f %.6f" % (u, v))
        self.nt += 1
        return self.nt

    def face(self, corners):
        """corners: list of (v, vt|None, vn|None); 3 or 4 of them."""
        toks = []
        for v, vt, vn in corners:
            if vt is not None and vn is not None:
                toks.append("%d/%d/%d" % (v, vt, vn))
            elif vn is not None:
                toks.append("%d//%d" % (v, vn))
            elif vt is not None:
                toks.append("%d/%d" % (v, vt))
            else:
                toks.append("%d" % v)
        line = "f " + " ".join(toks) + " "  # trailing space is REQUIRED
        assert len(line) < 126
        self.lines.append(line)
        self.ntris += len(corners) - 2

    def text(self) -> str:
        return "\n".join(self.lines) + "\n"


def add_grid(w: ObjWriter, origin, du, dv, nu, nv, normal, with_uv=False):
    """A tessellated parallelogram of nu x nv quads with one shared normal."""
    n = w.vn(*normal)
    idx = []
    uv = []
    for j in range(nv + 1):
        row = []
        urow = []
        for i in range(nu + 1):
            fu = i / nu
            fv = j / nv
            row.append(w.v(origin[0] + du[0] * fu + dv[0] * fv,
                           origin[1] + du[1] * fu + dv[1] * fv,
                           origin[2] + du[2] * fu + dv[2] * fv))
            if with_uv:
                urow.append(w.vt(fu * 4.0, fv * 4.0))
        idx.append(row)
        uv.append(urow)
    for j in range(nv):
        for i in range(nu):
            q = [(j, i), (j, i + 1), (j + 1, i + 1), (j + 1, i)]
            w.face([(idx[a][b], uv[a][b] if with_uv else None, n) for a, b in q])


def add_box(w: ObjWriter, lo, hi, div):
    """Axis-aligned box, outward normals, each face div x div quads."""
    x0, y0, z0 = lo
    x1, y1, z1 = hi
    dx, dy, dz = x1 - x0, y1 - y0, z1 - z0
    add_grid(w, (x0, y0, z0), (dx, 0, 0), (0, dy, 0), div, div, (0, 0, -1))
    add_grid(w, (x0, y0, z1), (dx, 0, 0), (0, dy, 0), div, div, (0, 0, 1))
    add_grid(w, (x0, y0, z0), (0, 0, dz), (0, dy, 0), div, div, (-1, 0, 0))
    add_grid(w, (x1, y0, z0), (0, 0, dz), (0, dy, 0), div, div, (1, 0, 0))
    add_grid(w, (x0, y0, z0), (dx, 0, 0), (0, 0, dz), div, div, (0, -1, 0))
    add_grid(w, (x0, y1, z0), (dx, 0, 0), (0, 0, dz), div, div, (0, 1, 0))


def add_sphere(w: ObjWriter, center, radius, level, smooth=True):
    verts, faces = icosphere(level)
    vi = []
    ni = []
    for x, y, z in verts:
        vi.append(w.v(center[0] + radius * x, center[1] + radius * y,
                      center[2] + radius * z))
        if smooth:
            ni.append(w.vn(x, y, z))
    for a, b, c in faces:
        if smooth:
            w.face([(vi[a], None, ni[a]), (vi[b], None, ni[b]), (vi[c], None, ni[c])])
        else:
            w.face([(vi[a], None, None), (vi[b], None, None), (vi[c], None, None)])


ROOM_MTL = """# synthetic room materials
newmtl white
Ka 0.75 0.75 0.75
Kd 0.75 0.75 0.75
Ks 0.1 0.1 0.1
Ns 10
newmtl red
Ka 0.75 0.15 0.15
Kd 0.75 0.15 0.15
Ks 0.0 0.0 0.0
Ns 1
newmtl green
Ka 0.15 0.75 0.15
Kd 0.15 0.75 0.15
Ks 0.0 0.0 0.0
Ns 1
newmtl blue
Ka 0.2 0.3 0.8
Kd 0.2 0.3 0.8
Ks 0.4 0.4 0.4
Ns 20
newmtl wood
Ka 0.55 0.35 0.2
Kd 0.55 0.35 0.2
Ks 0.05 0.05 0.05
Ns 5
newmtl floor
Ka 0.6 0.6 0.55
Kd 0.6 0.6 0.55
Ks 0.2 0.2 0.2
Ns 30
Refl 0.2
newmtl mirror
Ka 0.1 0.1 0.1
Kd 0.1 0.1 0.1
Ks 1.0 1.0 1.0
Ns 50
Refl 0.6
newmtl glass
Ka 0.05 0.05 0.05
Kd 0.05 0.05 0.05
Ks 0.8 0.8 0.8
Ns 80
Tr 0.8
Tf 0.9 0.95 1.0
Ni 1.5
\tillum 2
d 0.2
"""

ROOM_CAMERA = (200.0, 120.0, 20.0, 10.0, 0.0, 0.0, 110.0)
ROOM_LIGHTS = [
    (200.0, 230.0, 200.0, 0.3, 0.3, 0.3, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0),
    (60.0, 200.0, 60.0, 0.0, 0.0, 0.0, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3),
    (340.0, 200.0, 340.0, 0.0, 0.0, 0.0, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3),
]


def texture_ppm(kind: str, size: int = 128) -> bytes:
    """A procedural RGB texture as a binary PPM (P6), integer arithmetic only: "planks" (floor boards with a grain),
    "plaster" (a pale wall with speckles), "tiles" (two-tone checker with dark joints)."""
    rnd = SplitMix64({"planks": 11, "plaster": 12, "tiles": 13}[kind])
    px = bytearray()
    for y in range(size):
        for x in range(size):
            n = rnd.next() & 31
            if kind == "planks":
                board = (x * 8) // size
                joint = (x * 8) % size < 2 or (y + board * 37) % (size // 2) < 1
                g = 150 + ((board * 53) % 40) + (((y * 7 + board * 13) % 23) - 11) + (n >> 2)
                c = (40, 28, 20) if joint else (min(g + 40, 255), g, max(g - 60, 0))
            elif kind == "plaster":
                g = 215 + (n >> 1) - ((x ^ y) & 7)
                c = (g, g, max(g - 12, 0))
            else:
                t = ((x * 4) // size + (y * 4) // size) & 1
                joint = (x * 4) % size < 2 or (y * 4) % size < 2
                c = (60, 60, 60) if joint else ((225, 215, 190) if t else (120, 140, 170))
            px += bytes(c)
    return b"P6\n%d %d\n255\n" % (size, size) + bytes(px)


def room_text(name: str, wall_div: int, spheres, n_boxes: int, box_div: int,
              with_materials: bool, seed: int = 2017, textured: bool = False, clusters=()):
    """Returns (obj_text, mtl_text|None, n_triangles).

    spheres: list of (icosphere level, count).  textured: floor, ceiling and walls carry texture coordinates and
    map_Ka materials (the textures are written by write_scene).  clusters: list of (count, icosphere level, smallest
    radius, largest radius) -- tight groups of small spheres on a few shelves: the fine detail that makes the
    reference's octree deep."""
    w = ObjWriter((name + ".mtl") if with_materials else None)
    W, H, D = 400.0, 250.0, 400.0
    rnd = SplitMix64(seed)
    uv = textured
    # walls, normals facing inwards
    w.usemtl("floor")
    add_grid(w, (0, 0, 0), (W, 0, 0), (0, 0, D), wall_div, wall_div, (0, 1, 0), with_uv=uv)
    w.usemtl("white")
    add_grid(w, (0, H, 0), (W, 0, 0), (0, 0, D), wall_div, wall_div, (0, -1, 0), with_uv=uv)
    add_grid(w, (0, 0, D), (W, 0, 0), (0, H, 0), wall_div, wall_div, (0, 0, -1), with_uv=uv)
    add_grid(w, (0, 0, 0), (W, 0, 0), (0, H, 0), wall_div, wall_div, (0, 0, 1), with_uv=uv)
    w.usemtl("red")
    add_grid(w, (0, 0, 0), (0, 0, D), (0, H, 0), wall_div, wall_div, (1, 0, 0), with_uv=uv)
    w.usemtl("green")
    add_grid(w, (W, 0, 0), (0, 0, D), (0, H, 0), wall_div, wall_div, (-1, 0, 0), with_uv=uv)
    # a mirror panel slightly in front of the back wall
    w.usemtl("mirror")
    add_grid(w, (120, 60, D - 2.0), (160, 0, 0), (0, 120, 0), 4, 4, (0, 0, -1))
    sphere_mtls = ["blue", "glass", "mirror", "wood", "red", "white", "glass", "green"]
    k = 0
    for level, count in spheres:
        for _ in range(count):
            r = rnd.rng(8.0, 30.0)
            cx = rnd.rng(40.0, W - 40.0)
            cz = rnd.rng(90.0, D - 40.0)
            cy = r + rnd.rng(0.0, 120.0) * (1.0 if (k % 3) else 0.0)
            w.usemtl(sphere_mtls[k % len(sphere_mtls)])
            add_sphere(w, (cx, cy, cz), r, level)
            k += 1
    box_mtls = ["wood", "white", "blue", "wood", "glass"]
    for b in range(n_boxes):
        sx = rnd.rng(15.0, 60.0)
        sy = rnd.rng(10.0, 70.0)
        sz = rnd.rng(15.0, 60.0)
        x0 = rnd.rng(10.0, W - 10.0 - sx)
        z0 = rnd.rng(80.0, D - 10.0 - sz)
        w.usemtl(box_mtls[b % len(box_mtls)])
        add_box(w, (x0, 0.5, z0), (x0 + sx, 0.5 + sy, z0 + sz), box_div)
    # shelves of small things: every cluster sits in a box of 6 x 3 x 6 units somewhere in view
    for c, (count, level, r_lo, r_hi) in enumerate(clusters):
        bx = rnd.rng(60.0, W - 60.0)
        by = rnd.rng(20.0, 140.0)
        bz = rnd.rng(110.0, D - 60.0)
        for i in range(count):
            r = rnd.rng(r_lo, r_hi)
            w.usemtl(sphere_mtls[(c + i) % len(sphere_mtls)])
            add_sphere(w, (bx + rnd.rng(0.0, 6.0), by + rnd.rng(0.0, 3.0), bz + rnd.rng(0.0, 6.0)), r, level)
    mtl = ROOM_MTL if with_materials else None
    if mtl is not None and textured:
        for mname, tex in (("floor", "planks"), ("white", "plaster"), ("green", "tiles")):
            mtl = mtl.replace("newmtl %s\n" % mname, "newmtl %s\nmap_Ka %s_%s.ppm\n" % (mname, name, tex))
    return w.text(), mtl, w.ntris


# name -> generator arguments.  "room" is the BASELINE-sized scene.
SCENES = {
    "room":       dict(wall_div=40, spheres=[(3, 12), (4, 10)], n_boxes=20, box_div=8, with_materials=True),
    "room_nomtl": dict(wall_div=40, spheres=[(3, 12), (4, 10)], n_boxes=20, box_div=8, with_materials=False),
    "mini":       dict(wall_div=6, spheres=[(1, 3), (2, 4)], n_boxes=3, box_div=2, with_materials=True),
    "mini_nomtl": dict(wall_div=6, spheres=[(1, 3), (2, 4)], n_boxes=3, box_div=2, with_materials=False),
    # round 4, performance off the scene the kernels were tuned on (profiles/README.md):
    # the same room with map_Ka textures on floor, ceiling and walls: Texture::GetColorAt on the timed path
    "room_tex":   dict(wall_div=40, spheres=[(3, 12), (4, 10)], n_boxes=20, box_div=8, with_materials=True, textured=True),
    # 770 720 triangles, an octree of 16 levels (153 977 nodes): bigger and finer objects, and shelves of small spheres
    # (radius 0.03 .. 0.6) -- the deepest tree the hit-set walk takes (kHsMaxDepth)
    "loft":       dict(wall_div=48, spheres=[(4, 14), (5, 12)], n_boxes=24, box_div=10, with_materials=True,
                       clusters=[(60, 3, 0.04, 0.6), (60, 3, 0.04, 0.6), (40, 4, 0.1, 0.6), (30, 3, 0.03, 0.3)]),
    # the loft with ten to sixteen times finer things on the shelves (radius from 0.002): an octree of 20 levels, 154 793
    # nodes -- beyond round 3's walk (16 levels), inside the deep layout's (24)
    "loft_fine":  dict(wall_div=48, spheres=[(4, 14), (5, 12)], n_boxes=24, box_div=10, with_materials=True,
                       clusters=[(60, 3, 0.0025, 0.6), (60, 3, 0.005, 0.6), (40, 4, 0.006, 0.6), (30, 3, 0.002, 0.3)]),
}


def write_scene(name: str, out_dir: str) -> dict:
    """Writes <out_dir>/<name>.obj (+ .mtl).  Returns paths, count, sha256."""
    os.makedirs(out_dir, exist_ok=True)
    obj, mtl, ntris = room_text(name, **SCENES[name])
    obj_path = os.path.join(out_dir, name + ".obj")
    with open(obj_path, "w", newline="\n") as f:
        f.write(obj)
    if mtl is not None:
        with open(os.path.join(out_dir, name + ".mtl"), "w", newline="\n") as f:
            f.write(mtl)
        for tex in ("planks", "plaster", "tiles"):
            if ("%s_%s.ppm" % (name, tex)) in mtl:
                with open(os.path.join(out_dir, "%s_%s.ppm" % (name, tex)), "wb") as f:
                    f.write(texture_ppm(tex))
    return {"obj": obj_path, "triangles": ntris,
            "sha256": hashlib.sha256(obj.encode()).hexdigest()}


if __name__ == "__main__":
    import sys
    for n in sys.argv[2:] or list(SCENES):
        print(n, write_scene(n, sys.argv[1] if len(sys.argv) > 1 else "."))

"""ctypes binding of oracle/liboracle.so — the CPU oracle (test infrastructure).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
this module.  The product package never imports it.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")
REF_DRIVER = os.path.join(ORACLE_DIR, "_ref", "ref_driver")

CNT_NAMES = ["rays_primary", "rays_secondary", "rays_shadow", "box_tests",
             "node_visits", "tri_tests", "mt_tests", "shaded_hits"]

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liboracle.so"])


def lib():
    global _lib
    if _lib is not None:
        return _lib
    src = os.path.join(ORACLE_DIR, "mt_oracle.c")
    if (not os.path.exists(LIB_PATH)
            or os.path.getmtime(LIB_PATH) < os.path.getmtime(src)):
        build()
    L = ctypes.CDLL(LIB_PATH)
    vp, ci, cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
    L.orc_scene_new.restype = vp
    L.orc_scene_free.argtypes = [vp]
    L.orc_load_obj.argtypes = [vp, ctypes.c_char_p]
    L.orc_add_material.argtypes = [vp, ctypes.c_char_p, vp, vp, vp, cd, cd, cd, vp, cd]
    L.orc_add_texture.argtypes = [vp, ctypes.c_char_p, ci, ci, vp]
    L.orc_material_set_texture.argtypes = [vp, ci, ci]
    L.orc_add_triangle.argtypes = [vp, vp, vp, vp, ci, ci]
    L.orc_finalize.argtypes = [vp]
    L.orc_set_lights.argtypes = [vp, vp, ci]
    L.orc_num_triangles.argtypes = [vp]
    L.orc_num_materials.argtypes = [vp]
    L.orc_root_aabb.argtypes = [vp, vp]
    L.orc_get_triangle.argtypes = [vp, ci, vp, vp, vp]
    L.orc_get_material.argtypes = [vp, ci, vp, vp, vp]
    L.orc_tree_info.argtypes = [vp, vp, vp]
    L.orc_tree_dump.argtypes = [vp] * 7
    L.orc_sensor.argtypes = [vp, ci, ci, vp]
    L.orc_sensor_ray.argtypes = [vp, ci, ci, vp]
    L.orc_render_chunk.argtypes = [vp, vp] + [ci] * 7 + [vp] * 4 + [ci, vp]
    L.orc_intersect_rays.argtypes = [vp, ci] + [vp] * 8
    L.orc_tex_color_at.argtypes = [vp, ci, cd, cd, vp]
    L.orc_v3d_to_rgb.argtypes = [vp, vp]
    L.orc_last_error.restype = ctypes.c_char_p
    _lib = L
    return L


def _p(a):
    return None if a is None else a.ctypes.data


def _f64(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64))


class OracleScene:
    """The reference's Scene + OctTree + MythTracer, restated on the CPU."""

    def __init__(self, obj_path: str | None = None):
        self.L = lib()
        self.h = self.L.orc_scene_new()
        if obj_path is not None and not self.load_obj(obj_path):
            raise RuntimeError("oracle LoadObj failed: %s" %
                               self.L.orc_last_error().decode())

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_scene_free(self.h)
            self.h = None

    def load_obj(self, path: str) -> bool:
        return bool(self.L.orc_load_obj(self.h, os.fsencode(path)))

    def add_material(self, name, ka, kd, ks, ns=0.0, refl=0.0, tr=0.0,
                     tf=(0, 0, 0), ni=0.0) -> int:
        # arrays bound to names: a temporary would be freed before the call
        ka, kd, ks, tf = (_f64(ka).reshape(3), _f64(kd).reshape(3), _f64(ks).reshape(3),
                          _f64(tf).reshape(3))
        return self.L.orc_add_material(self.h, name.encode(), _p(ka), _p(kd), _p(ks),
                                       float(ns), float(refl), float(tr), _p(tf), float(ni))

    def add_texture(self, name, rgb) -> int:
        rgb = _f64(rgb)
        h, w, _ = rgb.shape
        return self.L.orc_add_texture(self.h, name.encode(), w, h, _p(rgb))

    def set_material_texture(self, mtl, tex):
        assert self.L.orc_material_set_texture(self.h, mtl, tex)

    def add_triangle(self, v, n=None, uvw=None, mtl=-1, line_no=0) -> int:
        v = _f64(v).reshape(9)
        n = None if n is None else _f64(n).reshape(9)
        uvw = None if uvw is None else _f64(uvw).reshape(9)
        return self.L.orc_add_triangle(self.h, _p(v), _p(n), _p(uvw), mtl, line_no)

    def finalize(self):
        if not self.L.orc_finalize(self.h):
            raise RuntimeError(self.L.orc_last_error().decode())

    def set_lights(self, lights):
        l = _f64(lights).reshape(-1, 12)
        self.L.orc_set_lights(self.h, _p(l), l.shape[0])

    @property
    def num_triangles(self):
        return self.L.orc_num_triangles(self.h)

    @property
    def num_materials(self):
        return self.L.orc_num_materials(self.h)

    def root_aabb(self):
        o = np.zeros(6)
        self.L.orc_root_aabb(self.h, _p(o))
        return o

    def triangles(self):
        n = self.num_triangles
        data = np.zeros((n, 33))
        mtl = np.zeros(n, dtype=np.int32)
        line = np.zeros(n, dtype=np.int32)
        m, l = ctypes.c_int(), ctypes.c_int()
        for i in range(n):
            self.L.orc_get_triangle(self.h, i, data[i].ctypes.data,
                                    ctypes.addressof(m), ctypes.addressof(l))
            mtl[i], line[i] = m.value, l.value
        return data, mtl, line

    def materials(self):
        out = []
        for i in range(self.num_materials):
            d = np.zeros(16)
            tex = ctypes.c_int()
            name = ctypes.create_string_buffer(128)
            self.L.orc_get_material(self.h, i, _p(d), ctypes.addressof(tex), name)
            out.append((name.value.decode(), d, tex.value))
        return out

    def tree(self):
        self.finalize()
        nn, dd = ctypes.c_int(), ctypes.c_int()
        self.L.orc_tree_info(self.h, ctypes.addressof(nn), ctypes.addressof(dd))
        n = nn.value
        t = dict(depth=dd.value,
                 aabb=np.zeros((n, 6)), center=np.zeros((n, 3)),
                 first_child=np.zeros(n, dtype=np.int32),
                 prim_begin=np.zeros(n, dtype=np.int32),
                 prim_count=np.zeros(n, dtype=np.int32),
                 prim_ids=np.zeros(max(self.num_triangles, 1), dtype=np.int32))
        self.L.orc_tree_dump(self.h, _p(t["aabb"]), _p(t["center"]),
                             _p(t["first_child"]), _p(t["prim_begin"]),
                             _p(t["prim_count"]), _p(t["prim_ids"]))
        t["prim_ids"] = t["prim_ids"][:self.num_triangles]
        return t

    def render(self, cam, image_w, image_h, chunk=None, max_level=5,
               debug=False, nthreads=0):
        cx, cy, cw, ch = chunk if chunk else (0, 0, image_w, image_h)
        cam = _f64(cam)
        rgb = np.zeros((ch, cw, 3), dtype=np.uint8)
        dl = np.zeros((ch, cw), dtype=np.int32) if debug else None
        dp = np.zeros((ch, cw, 3)) if debug else None
        cnt = np.zeros(len(CNT_NAMES), dtype=np.uint64)
        sec = ctypes.c_double()
        ok = self.L.orc_render_chunk(self.h, _p(cam), image_w, image_h, cx, cy,
                                     cw, ch, max_level, _p(rgb), _p(dl), _p(dp),
                                     _p(cnt), nthreads, ctypes.addressof(sec))
        if not ok:
            raise RuntimeError(self.L.orc_last_error().decode())
        return dict(rgb=rgb, line=dl, point=dp, seconds=sec.value,
                    counters=dict(zip(CNT_NAMES, (int(c) for c in cnt))))

    def intersect(self, rays):
        rays = _f64(rays).reshape(-1, 6)
        n = rays.shape[0]
        out = dict(tri=np.zeros(n, dtype=np.int32), line=np.zeros(n, dtype=np.int32),
                   t=np.zeros(n), point=np.zeros((n, 3)), normal=np.zeros((n, 3)),
                   uvw=np.zeros((n, 3)))
        cnt = np.zeros(len(CNT_NAMES), dtype=np.uint64)
        self.L.orc_intersect_rays(self.h, n, _p(rays), _p(out["tri"]), _p(out["line"]),
                                  _p(out["t"]), _p(out["point"]), _p(out["normal"]),
                                  _p(out["uvw"]), _p(cnt))
        out["counters"] = dict(zip(CNT_NAMES, (int(c) for c in cnt)))
        return out

    def tex_color_at(self, tex, u, v):
        o = np.zeros(3)
        self.L.orc_tex_color_at(self.h, tex, float(u), float(v), _p(o))
        return o


def sensor(cam, w, h):
    o = np.zeros(12)
    cam = _f64(cam)
    lib().orc_sensor(_p(cam), w, h, _p(o))
    return o


def sensor_ray(sens, x, y):
    d = np.zeros(3)
    sens = _f64(sens)
    lib().orc_sensor_ray(_p(sens), x, y, _p(d))
    return d


def v3d_to_rgb(v):
    o = np.zeros(3, dtype=np.uint8)
    v = _f64(v)
    lib().orc_v3d_to_rgb(_p(v), _p(o))
    return o


# ---------------------------------------------------------------------------
# The real reference (oracle/_ref/ref_driver), when it has been built.

def have_ref() -> bool:
    return os.path.exists(REF_DRIVER)


DEBUG_DTYPE = np.dtype([("line", "<i4"), ("point", "<f8", 3)])
RAYOUT_DTYPE = np.dtype([("line", "<i4"), ("t", "<f8"), ("point", "<f8", 3),
                         ("normal", "<f8", 3), ("uvw", "<f8", 3)])


def run_ref(workdir, obj, image=None, chunk=None, cam=None, lights=(),
            want_rgb=True, want_debug=False, rays=None, want_sensor=False,
            repeat=1, threads=None):
    """Runs one job through the compiled reference; returns a dict."""
    os.makedirs(workdir, exist_ok=True)
    job = ["obj %s" % obj]
    W, H = image if image else (0, 0)
    job.append("image %d %d" % (W, H))
    if chunk:
        job.append("chunk %d %d %d %d" % tuple(chunk))
    if cam is not None:
        job.append("camera " + " ".join(repr(float(c)) for c in cam))
    for l in lights:
        job.append("light " + " ".join(repr(float(c)) for c in l))
    out = {}
    if image and want_rgb:
        job.append("out_rgb %s" % os.path.join(workdir, "out.raw"))
        job.append("out_time %s" % os.path.join(workdir, "time.json"))
        job.append("repeat %d" % repeat)
    if image and want_debug:
        job.append("out_debug %s" % os.path.join(workdir, "out.dbg"))
    if rays is not None:
        _f64(rays).reshape(-1, 6).tofile(os.path.join(workdir, "rays.bin"))
        job.append("rays %s %s" % (os.path.join(workdir, "rays.bin"),
                                   os.path.join(workdir, "rays.out")))
    if image and want_sensor:
        job.append("sensor %s" % os.path.join(workdir, "sensor.bin"))
    jpath = os.path.join(workdir, "job.txt")
    with open(jpath, "w") as f:
        f.write("\n".join(job) + "\n")
    env = dict(os.environ)
    if threads:
        env["OMP_NUM_THREADS"] = str(threads)
    r = subprocess.run([REF_DRIVER, jpath], env=env, stderr=subprocess.PIPE)
    out["returncode"] = r.returncode
    out["stderr"] = r.stderr.decode(errors="replace")
    if r.returncode != 0:
        return out
    cw, ch = (chunk[2], chunk[3]) if chunk else (W, H)
    if image and want_rgb:
        out["rgb"] = np.fromfile(os.path.join(workdir, "out.raw"),
                                 dtype=np.uint8).reshape(ch, cw, 3)
        import json
        out["time"] = json.load(open(os.path.join(workdir, "time.json")))
    if image and want_debug:
        d = np.fromfile(os.path.join(workdir, "out.dbg"), dtype=DEBUG_DTYPE)
        out["line"] = d["line"].reshape(ch, cw)
        out["point"] = d["point"].reshape(ch, cw, 3)
    if rays is not None:
        out["rays"] = np.fromfile(os.path.join(workdir, "rays.out"), dtype=RAYOUT_DTYPE)
    if image and want_sensor:
        out["sensor"] = np.fromfile(os.path.join(workdir, "sensor.bin")).reshape(ch, cw, 3)
    return out

"""ctypes bindings of the two native libraries (see package docstring)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
HIP_LIB = os.path.join(PKG, "lib", "libmythtracer_hip.so")
HOST_LIB = os.path.join(PKG, "lib", "libmythtracer_host.so")

MT_OK = 0
MT_ABI_VERSION = 4
MT_TEX_RGB8, MT_TEX_F64 = 0, 1

# every symbol include/mythtracer_hip.h declares
HIP_SYMBOLS = [
    "mt_last_error", "mt_abi_version", "mt_device_count", "mt_scene_create",
    "mt_scene_destroy", "mt_scene_set_lights", "mt_render_chunk",
    "mt_render_chunk_device", "mt_render_tiles_device", "mt_blit_tiles_device",
    "mt_scene_read_stats", "mt_intersect_rays", "mt_scene_set_traversal_mode",
    "mt_scene_kernel_times", "mt_scene_set_scheduling", "mt_scene_set_engine",
    "mt_scene_set_stats", "mt_set_default_engine", "mt_scene_set_tuning",
    "mt_render_frame_multi", "mt_scene_export_costs_device", "mt_scene_import_costs_device",
    "mt_order_tiles_device", "mt_dealt_tile_count", "mt_deal_tiles_device",
    "mt_render_tile_list_device", "mt_blit_tile_list_device",
]

# mt_scene_set_tuning knobs, in the order of the enum in include/mythtracer_hip.h
TUNE = {name: i for i, name in enumerate([
    "POOL_BELOW", "POOL_CAP", "PACKED_STACK", "BLOCKS_PER_CU", "FORECAST_RADIUS", "BLEND", "FORMS",
    "POOL_CUT_SHARE", "POOL_PIECE_TIME1", "POOL_PIECE_TIME2", "POOL_PIECE_WORK1", "POOL_PIECE_WORK2",
    "POOL_CELL_FACTOR", "QUAD_SHARE", "QUAD_SHARE_MOVING", "QUAD_KEEP", "QUAD_WORK", "QUAD_WORK_MOVING",
    "POOL_SCRATCH_MB", "HYBRID_POOL_SHARE", "HYBRID_QUAD_SHARE", "HYBRID_WORK1", "HYBRID_WORK2", "FORECAST_STEP",
    "HYBRID_STARTER_SHARE", "DEEP_LAYOUT", "MULTI_FORCE_PEER_COPY", "MULTI_BALANCE", "XCD_QUEUES",
    "ORDER_GROUPS", "SM_CELL_SHARE", "SM_CELL_TIME", "SM_CELL_WORK", "HYBRID_CELL_FACTOR"])}

STAT_NAMES = ["rays_primary", "rays_secondary", "rays_shadow", "box_tests",
              "node_visits", "tri_tests", "mt_tests", "shaded_hits"]


class NativeLibraryMissing(RuntimeError):
    pass


class mt_material(C.Structure):
    _fields_ = [("ambient", C.c_double * 3), ("diffuse", C.c_double * 3),
                ("specular", C.c_double * 3), ("specular_exp", C.c_double),
                ("reflectance", C.c_double), ("transparency", C.c_double),
                ("transmission_filter", C.c_double * 3),
                ("refraction_index", C.c_double), ("tex", C.c_int32),
                ("reserved", C.c_int32)]


class mt_light(C.Structure):
    _fields_ = [("position", C.c_double * 3), ("ambient", C.c_double * 3),
                ("diffuse", C.c_double * 3), ("specular", C.c_double * 3)]


class mt_texture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("format", C.c_int32),
                ("reserved", C.c_int32), ("texels", C.c_void_p)]


class mt_sensor(C.Structure):
    _fields_ = [("origin", C.c_double * 3), ("start_point", C.c_double * 3),
                ("delta_scanline", C.c_double * 3), ("delta_pixel", C.c_double * 3)]


class mt_stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in STAT_NAMES] + [
        ("wave_node_steps", C.c_uint64), ("wave_tri_steps", C.c_uint64),
        ("kernel_ms", C.c_double), ("total_ms", C.c_double),
        ("bytes_scalar", C.c_uint64), ("bytes_vector", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class mt_scene_desc(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("abi_version", C.c_uint32),
                ("device", C.c_int32), ("n_nodes", C.c_int32), ("n_tris", C.c_int32),
                ("n_materials", C.c_int32), ("n_textures", C.c_int32),
                ("tree_depth", C.c_int32),
                ("node_aabb", C.c_void_p), ("node_center", C.c_void_p),
                ("node_first_child", C.c_void_p), ("node_prim_begin", C.c_void_p),
                ("node_prim_count", C.c_void_p),
                ("tri_vertex", C.c_void_p), ("tri_normal", C.c_void_p),
                ("tri_uvw", C.c_void_p), ("tri_aabb", C.c_void_p),
                ("tri_material", C.c_void_p), ("tri_line_no", C.c_void_p),
                ("tri_id", C.c_void_p),
                ("materials", C.c_void_p), ("textures", C.c_void_p)]


DEBUG_PX_DTYPE = np.dtype([("line_no", "<i4"), ("reserved", "<i4"), ("point", "<f8", 3)])


def _load(path):
    if not os.path.exists(path):
        raise NativeLibraryMissing(
            "%s is missing: run `python -m mythtracer_amd.build` (needs hipcc); "
            "there is no CPU fallback" % path)
    return C.CDLL(path)


def _ptr(a):
    return None if a is None else a.ctypes.data


def _f64(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64))


def _i32(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.int32))


class HipAbi:
    """libmythtracer_hip.so — the C ABI, one method per entry point."""

    def __init__(self, lib_path=None):
        L = self.lib = _load(lib_path or HIP_LIB)
        vp, ci = C.c_void_p, C.c_int
        L.mt_last_error.restype = C.c_char_p
        L.mt_scene_create.restype = vp
        L.mt_scene_create.argtypes = [C.POINTER(mt_scene_desc)]
        L.mt_scene_destroy.argtypes = [vp]
        L.mt_scene_destroy.restype = None
        L.mt_scene_set_lights.argtypes = [vp, vp, ci]
        L.mt_render_chunk.argtypes = [vp, C.POINTER(mt_sensor)] + [ci] * 7 + [vp, vp, vp]
        L.mt_render_chunk_device.argtypes = [vp, C.POINTER(mt_sensor)] + [ci] * 7 + [vp, vp, vp]
        L.mt_render_tiles_device.argtypes = [vp, C.POINTER(mt_sensor)] + [ci] * 8 + [vp, vp]
        L.mt_blit_tiles_device.argtypes = [vp] + [ci] * 7 + [vp, vp, vp]
        L.mt_scene_read_stats.argtypes = [vp, C.POINTER(mt_stats)]
        L.mt_intersect_rays.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp]
        L.mt_scene_set_traversal_mode.argtypes = [vp, ci]
        L.mt_scene_kernel_times.argtypes = [vp, ci, vp, vp]
        L.mt_scene_set_scheduling.argtypes = [vp, ci]
        L.mt_scene_set_engine.argtypes = [vp, ci]
        L.mt_scene_set_stats.argtypes = [vp, ci]
        L.mt_set_default_engine.argtypes = [ci]
        L.mt_scene_set_tuning.argtypes = [vp, ci, C.c_double]
        L.mt_render_frame_multi.argtypes = [vp, ci, C.POINTER(mt_sensor)] + [ci] * 5 + [vp, vp]
        L.mt_scene_export_costs_device.argtypes = [vp, vp, ci, ci, vp]
        L.mt_scene_import_costs_device.argtypes = [vp, vp, ci, ci, vp]
        L.mt_order_tiles_device.argtypes = [vp, vp] + [ci] * 6 + [vp, vp]
        L.mt_dealt_tile_count.argtypes = [ci] * 6
        L.mt_deal_tiles_device.argtypes = [vp, vp] + [ci] * 6 + [vp, vp]
        L.mt_render_tile_list_device.argtypes = [vp, C.POINTER(mt_sensor)] + [ci] * 4 + [vp, ci, C.c_uint64, ci, vp, vp]
        L.mt_blit_tile_list_device.argtypes = [vp] + [ci] * 4 + [vp, ci, vp, vp, vp]

    def last_error(self) -> str:
        return self.lib.mt_last_error().decode(errors="replace")

    def check(self, rc):
        if rc != MT_OK:
            raise RuntimeError("mythtracer_hip error %d: %s" % (rc, self.last_error()))

    def device_count(self) -> int:
        return self.lib.mt_device_count()

    @staticmethod
    def make_sensor(sensor12) -> mt_sensor:
        s = _f64(sensor12).reshape(12)
        out = mt_sensor()
        for i, name in enumerate(["origin", "start_point", "delta_scanline", "delta_pixel"]):
            for k in range(3):
                getattr(out, name)[k] = s[i * 3 + k]
        return out

    def scene_create(self, flat: dict, device: int = 0):
        """flat: dict of numpy arrays in the layout of mt_scene_desc (see
        MythTracer.flatten()).  Returns an opaque handle."""
        keep = []

        def arr(name, dt):
            a = np.ascontiguousarray(flat[name], dtype=dt)
            keep.append(a)
            return a.ctypes.data

        d = mt_scene_desc()
        d.struct_size = C.sizeof(mt_scene_desc)
        d.abi_version = MT_ABI_VERSION
        d.device = device
        d.n_nodes = len(flat["node_first_child"])
        d.n_tris = len(flat["tri_material"])
        d.tree_depth = int(flat["tree_depth"])
        d.node_aabb = arr("node_aabb", np.float64)
        d.node_center = arr("node_center", np.float64)
        d.node_first_child = arr("node_first_child", np.int32)
        d.node_prim_begin = arr("node_prim_begin", np.int32)
        d.node_prim_count = arr("node_prim_count", np.int32)
        d.tri_vertex = arr("tri_vertex", np.float64)
        d.tri_normal = arr("tri_normal", np.float64)
        d.tri_uvw = arr("tri_uvw", np.float64)
        d.tri_aabb = arr("tri_aabb", np.float64)
        d.tri_material = arr("tri_material", np.int32)
        d.tri_line_no = arr("tri_line_no", np.int32)
        d.tri_id = arr("tri_id", np.int32)
        mats = flat.get("materials", [])
        marr = (mt_material * max(len(mats), 1))()
        for i, m in enumerate(mats):
            v = _f64(m["values"])  # ka kd ks ns refl tr tf ni
            for k in range(3):
                marr[i].ambient[k] = v[k]
                marr[i].diffuse[k] = v[3 + k]
                marr[i].specular[k] = v[6 + k]
                marr[i].transmission_filter[k] = v[12 + k]
            marr[i].specular_exp, marr[i].reflectance, marr[i].transparency = v[9], v[10], v[11]
            marr[i].refraction_index = v[15]
            marr[i].tex = int(m.get("tex", -1))
        texs = flat.get("textures", [])
        tarr = (mt_texture * max(len(texs), 1))()
        for i, t in enumerate(texs):
            texels = np.ascontiguousarray(t["texels"])
            keep.append(texels)
            tarr[i].height, tarr[i].width = texels.shape[0], texels.shape[1]
            tarr[i].format = MT_TEX_RGB8 if texels.dtype == np.uint8 else MT_TEX_F64
            tarr[i].texels = texels.ctypes.data
        keep += [marr, tarr]
        d.n_materials = len(mats)
        d.n_textures = len(texs)
        d.materials = C.cast(marr, C.c_void_p)
        d.textures = C.cast(tarr, C.c_void_p)
        h = self.lib.mt_scene_create(C.byref(d))
        if not h:
            raise RuntimeError("mt_scene_create failed: " + self.last_error())
        return h

    def scene_destroy(self, h):
        self.lib.mt_scene_destroy(h)

    def set_lights(self, h, lights):
        l = _f64(lights).reshape(-1, 12)
        self.check(self.lib.mt_scene_set_lights(h, _ptr(l), l.shape[0]))

    def set_traversal_mode(self, h, mode: int):
        self.check(self.lib.mt_scene_set_traversal_mode(h, mode))

    def render_chunk(self, h, sensor12, image_w, image_h, chunk=None, max_depth=5, debug=False):
        cx, cy, cw, ch = chunk if chunk else (0, 0, image_w, image_h)
        rgb = np.zeros((max(ch, 0), max(cw, 0), 3), dtype=np.uint8)
        dbg = np.zeros((max(ch, 0), max(cw, 0)), dtype=DEBUG_PX_DTYPE) if debug else None
        st = mt_stats()
        s = self.make_sensor(sensor12)
        self.check(self.lib.mt_render_chunk(h, C.byref(s), image_w, image_h, cx, cy, cw, ch,
                                            max_depth, _ptr(rgb), _ptr(dbg), C.addressof(st)))
        out = dict(rgb=rgb, stats=st.as_dict())
        if debug:
            out["line"] = dbg["line_no"].copy()
            out["point"] = dbg["point"].copy()
        return out

    def render_chunk_device(self, h, sensor12, image_w, image_h, chunk, max_depth, d_rgb,
                            d_debug=None, stream=None):
        s = self.make_sensor(sensor12)
        self.check(self.lib.mt_render_chunk_device(h, C.byref(s), image_w, image_h, *chunk,
                                                   max_depth, d_rgb, d_debug, stream))

    def render_tiles_device(self, h, sensor12, image_w, image_h, tile_w, tile_h, first_tile,
                            tile_stride, n_tiles, max_depth, d_rgb, stream=None):
        s = self.make_sensor(sensor12)
        self.check(self.lib.mt_render_tiles_device(h, C.byref(s), image_w, image_h, tile_w, tile_h,
                                                   first_tile, tile_stride, n_tiles, max_depth,
                                                   d_rgb, stream))

    def blit_tiles_device(self, h, image_w, image_h, tile_w, tile_h, first_tile, tile_stride,
                          n_tiles, d_tiles, d_image, stream=None):
        self.check(self.lib.mt_blit_tiles_device(h, image_w, image_h, tile_w, tile_h, first_tile,
                                                 tile_stride, n_tiles, d_tiles, d_image, stream))

    # ---- cost-balanced tile ownership (include/mythtracer_hip.h, mt_order_tiles_device ff.)
    def order_tiles_device(self, h, d_cost_map, map_w, map_h, image_w, image_h, tile_w, tile_h, d_order, stream=None):
        self.check(self.lib.mt_order_tiles_device(h, d_cost_map, map_w, map_h, image_w, image_h, tile_w, tile_h,
                                                  d_order, stream))

    def dealt_tile_count(self, image_w, image_h, tile_w, tile_h, world, rank) -> int:
        n = self.lib.mt_dealt_tile_count(image_w, image_h, tile_w, tile_h, world, rank)
        if n < 0:
            self.check(n)
        return n

    def deal_tiles_device(self, h, d_order, image_w, image_h, tile_w, tile_h, world, rank, d_list, stream=None) -> int:
        n = self.lib.mt_deal_tiles_device(h, d_order, image_w, image_h, tile_w, tile_h, world, rank, d_list, stream)
        if n < 0:
            self.check(n)
        return n

    def render_tile_list_device(self, h, sensor12, image_w, image_h, tile_w, tile_h, d_list, n_tiles, list_id,
                                max_depth, d_rgb, stream=None):
        s = self.make_sensor(sensor12)
        self.check(self.lib.mt_render_tile_list_device(h, C.byref(s), image_w, image_h, tile_w, tile_h, d_list,
                                                       n_tiles, int(list_id), max_depth, d_rgb, stream))

    def blit_tile_list_device(self, h, image_w, image_h, tile_w, tile_h, d_list, n_tiles, d_tiles, d_image,
                              stream=None):
        self.check(self.lib.mt_blit_tile_list_device(h, image_w, image_h, tile_w, tile_h, d_list, n_tiles,
                                                     d_tiles, d_image, stream))

    def read_stats(self, h) -> dict:
        st = mt_stats()
        self.check(self.lib.mt_scene_read_stats(h, C.byref(st)))
        return st.as_dict()

    def set_stats(self, h, enabled: bool):
        """Work counters of the *_device calls on (default) or off."""
        self.check(self.lib.mt_scene_set_stats(h, 1 if enabled else 0))

    def set_engine(self, h, engine: int):
        """0 = automatic, 1 = throughput engine (state machine), 2 = latency engine (ray pool), 3 = hybrid."""
        self.check(self.lib.mt_scene_set_engine(h, int(engine)))

    def set_default_engine(self, engine: int):
        """Engine of the scenes created from now on (also those the C++ facade creates)."""
        self.check(self.lib.mt_set_default_engine(int(engine)))

    def set_tuning(self, h, knob: str, value: float):
        """mt_scene_set_tuning; knob = a key of TUNE (e.g. "POOL_CAP")."""
        self.check(self.lib.mt_scene_set_tuning(h, TUNE[knob], float(value)))

    def render_frame_multi(self, handles, sensor12, image_w, image_h, tile_w=64, tile_h=64, max_depth=5,
                           want_stats=True):
        """mt_render_frame_multi: one frame on the replicas `handles` (one per GPU)."""
        n = len(handles)
        arr = (C.c_void_p * n)(*handles)
        rgb = np.zeros((image_h, image_w, 3), dtype=np.uint8)
        st = (mt_stats * n)()
        s = self.make_sensor(sensor12)
        self.check(self.lib.mt_render_frame_multi(C.cast(arr, C.c_void_p), n, C.byref(s), image_w, image_h,
                                                  tile_w, tile_h, max_depth, _ptr(rgb),
                                                  C.cast(st, C.c_void_p) if want_stats else None))
        return dict(rgb=rgb, stats=[st[i].as_dict() for i in range(n)])

    def export_costs_device(self, h, d_map, map_w, map_h, stream=None):
        self.check(self.lib.mt_scene_export_costs_device(h, d_map, map_w, map_h, stream))

    def import_costs_device(self, h, d_map, map_w, map_h, stream=None):
        self.check(self.lib.mt_scene_import_costs_device(h, d_map, map_w, map_h, stream))

    def set_scheduling(self, h, use_cost_history: bool):
        self.check(self.lib.mt_scene_set_scheduling(h, 1 if use_cost_history else 0))

    def kernel_times(self, h, max_n: int = 64):
        """(primary_ms[], render_ms[]) of the launches since the previous call."""
        a = np.zeros(max_n)
        b = np.zeros(max_n)
        n = self.lib.mt_scene_kernel_times(h, max_n, a.ctypes.data_as(C.c_void_p),
                                           b.ctypes.data_as(C.c_void_p))
        if n < 0:
            self.check(n)
        return a[:n].copy(), b[:n].copy()

    def intersect_rays(self, h, rays):
        rays = _f64(rays).reshape(-1, 6)
        n = rays.shape[0]
        tri = np.zeros(n, dtype=np.int32)
        line = np.zeros(n, dtype=np.int32)
        t = np.zeros(n)
        point = np.zeros((n, 3))
        st = mt_stats()
        self.check(self.lib.mt_intersect_rays(h, n, _ptr(rays), _ptr(tri), _ptr(line), _ptr(t),
                                              _ptr(point), C.addressof(st)))
        return dict(tri=tri, line=line, t=t, point=point, stats=st.as_dict())


_hip = None
_host = None


def hip_abi() -> HipAbi:
    global _hip
    if _hip is None:
        _hip = HipAbi()
    return _hip


def host_lib():
    global _host
    if _host is not None:
        return _host
    L = _load(HOST_LIB)
    vp, ci, cd, cs = C.c_void_p, C.c_int, C.c_double, C.c_char_p
    L.mth_new.restype = vp
    L.mth_new.argtypes = [ci, ci]
    L.mth_free.argtypes = [vp]
    L.mth_free.restype = None
    L.mth_set_devices.argtypes = [vp, vp, ci]
    L.mth_set_devices.restype = None
    L.mth_last_error.restype = cs
    L.mth_last_error.argtypes = [vp]
    L.mth_load_obj.argtypes = [vp, cs]
    L.mth_add_material.argtypes = [vp, cs, vp, vp, vp, cd, cd, cd, vp, cd]
    L.mth_add_texture.argtypes = [vp, cs, ci, ci, vp]
    L.mth_material_set_texture.argtypes = [vp, ci, ci]
    L.mth_add_triangle.argtypes = [vp, vp, vp, vp, ci, ci]
    L.mth_set_lights.argtypes = [vp, vp, ci]
    L.mth_set_lights.restype = None
    L.mth_set_max_level.argtypes = [vp, ci]
    L.mth_set_max_level.restype = None
    L.mth_finalize.argtypes = [vp]
    L.mth_finalize.restype = None
    L.mth_prepare.argtypes = [vp]
    L.mth_device_scene.argtypes = [vp]
    L.mth_device_scene.restype = vp
    L.mth_root_aabb.argtypes = [vp, vp]
    L.mth_root_aabb.restype = None
    L.mth_tree_info.argtypes = [vp, vp, vp, vp]
    L.mth_tree_info.restype = None
    L.mth_tree_dump.argtypes = [vp] * 7
    L.mth_tree_dump.restype = None
    L.mth_triangles.argtypes = [vp, vp, vp, vp]
    L.mth_triangles.restype = None
    L.mth_flatten.argtypes = [vp] * 12
    L.mth_flatten_texels.argtypes = [vp, ci, vp]
    L.mth_num_materials.argtypes = [vp]
    L.mth_get_material.argtypes = [vp, cs, vp, vp]
    L.mth_sensor.argtypes = [vp, ci, ci, vp]
    L.mth_sensor.restype = None
    L.mth_sensor_ray.argtypes = [vp, ci, ci, ci, ci, vp]
    L.mth_sensor_ray.restype = None
    L.mth_render_chunk.argtypes = [vp, vp] + [ci] * 6 + [vp] * 5
    L.mth_render_image.argtypes = [vp, vp, ci, ci, vp]
    L.mth_frame_loop.argtypes = [vp, vp, ci, ci, ci, cd, ci, vp, vp]
    L.mth_intersect.argtypes = [vp, ci, vp, vp, vp, vp, vp]
    L.mth_chunk_serialize_input.argtypes = [vp, vp]
    L.mth_chunk_deserialize_input.argtypes = [vp, ci, vp]
    L.mth_chunk_output_roundtrip.argtypes = [ci, ci, vp, ci, vp, ci, vp]
    L.mth_chunk_deserialize_output.argtypes = [ci, ci, vp, ci]
    L.mth_camera_roundtrip.argtypes = [vp, vp, vp]
    _host = L
    return L


def sensor(cam7, w, h):
    """Camera::GetSensor on the host -> origin, start, delta_scanline, delta_pixel."""
    out = np.zeros(12)
    cam = _f64(cam7)  # bound to a name: must outlive the call
    host_lib().mth_sensor(_ptr(cam), w, h, _ptr(out))
    return out


def sensor_ray(cam7, w, h, x, y):
    d = np.zeros(3)
    cam = _f64(cam7)
    host_lib().mth_sensor_ray(_ptr(cam), w, h, x, y, _ptr(d))
    return d


class MythTracer:
    """raytracer::MythTracer (host facade) through the ctypes shim."""

    def __init__(self, obj_path=None, device=0, quiet=True):
        self.L = host_lib()
        self.h = self.L.mth_new(device, 1 if quiet else 0)
        if obj_path is not None and not self.load_obj(obj_path):
            raise RuntimeError("LoadObj failed for %s" % obj_path)

    def close(self):
        if getattr(self, "h", None):
            self.L.mth_free(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def last_error(self):
        return self.L.mth_last_error(self.h).decode(errors="replace")

    def set_devices(self, devices):
        """MythTracer::SetDevices: the W x H overload of RayTrace renders on all of them."""
        d = _i32(devices)
        self.L.mth_set_devices(self.h, _ptr(d), len(d))

    def load_obj(self, path) -> bool:
        return bool(self.L.mth_load_obj(self.h, os.fsencode(path)))

    def add_material(self, name, ka, kd, ks, ns=0.0, refl=0.0, tr=0.0, tf=(0, 0, 0), ni=0.0):
        # every converted array is bound to a name so that it outlives the call
        # (a temporary's buffer is recycled by numpy before the C side reads it)
        ka, kd, ks, tf = _f64(ka).reshape(3), _f64(kd).reshape(3), _f64(ks).reshape(3), _f64(tf).reshape(3)
        return self.L.mth_add_material(self.h, name.encode(), _ptr(ka), _ptr(kd), _ptr(ks),
                                       float(ns), float(refl), float(tr), _ptr(tf), float(ni))

    def add_texture(self, name, rgb):
        rgb = _f64(rgb)
        h, w, _ = rgb.shape
        return self.L.mth_add_texture(self.h, name.encode(), w, h, _ptr(rgb))

    def set_material_texture(self, mtl, tex):
        assert self.L.mth_material_set_texture(self.h, mtl, tex)

    def add_triangle(self, v, n=None, uvw=None, mtl=-1, line_no=0):
        v = _f64(v).reshape(9)
        n = None if n is None else _f64(n).reshape(9)
        uvw = None if uvw is None else _f64(uvw).reshape(9)
        return self.L.mth_add_triangle(self.h, _ptr(v), _ptr(n), _ptr(uvw), mtl, line_no)

    def set_lights(self, lights):
        l = _f64(lights).reshape(-1, 12)
        self.L.mth_set_lights(self.h, _ptr(l), l.shape[0])

    def set_max_level(self, level):
        self.L.mth_set_max_level(self.h, level)

    def finalize(self):
        self.L.mth_finalize(self.h)

    def prepare(self):
        if not self.L.mth_prepare(self.h):
            raise RuntimeError("MythTracer::Prepare failed: " + self.last_error())

    def device_scene(self):
        self.prepare()
        return self.L.mth_device_scene(self.h)

    def root_aabb(self):
        o = np.zeros(6)
        self.L.mth_root_aabb(self.h, _ptr(o))
        return o

    def tree(self):
        self.finalize()
        nn, nt, dd = C.c_int(), C.c_int(), C.c_int()
        self.L.mth_tree_info(self.h, C.addressof(nn), C.addressof(nt), C.addressof(dd))
        n = nn.value
        t = dict(depth=dd.value, aabb=np.zeros((n, 6)), center=np.zeros((n, 3)),
                 first_child=np.zeros(n, dtype=np.int32), prim_begin=np.zeros(n, dtype=np.int32),
                 prim_count=np.zeros(n, dtype=np.int32),
                 prim_ids=np.zeros(max(nt.value, 1), dtype=np.int32))
        self.L.mth_tree_dump(self.h, _ptr(t["aabb"]), _ptr(t["center"]), _ptr(t["first_child"]),
                             _ptr(t["prim_begin"]), _ptr(t["prim_count"]), _ptr(t["prim_ids"]))
        t["prim_ids"] = t["prim_ids"][:nt.value]
        return t

    def triangles(self):
        nn, nt, dd = C.c_int(), C.c_int(), C.c_int()
        self.L.mth_tree_info(self.h, C.addressof(nn), C.addressof(nt), C.addressof(dd))
        n = nt.value
        data = np.zeros((max(n, 1), 33))
        line = np.zeros(max(n, 1), dtype=np.int32)
        has = np.zeros(max(n, 1), dtype=np.int32)
        self.L.mth_triangles(self.h, _ptr(data), _ptr(line), _ptr(has))
        return data[:n], line[:n], has[:n]

    def flatten(self) -> dict:
        """The flattened scene exactly as MythTracer::Prepare passes it to
        mt_scene_create (input of HipAbi.scene_create)."""
        t = self.tree()
        nt, nm, nx = C.c_int(), C.c_int(), C.c_int()
        a = [C.addressof(nt), C.addressof(nm), C.addressof(nx)]
        if not self.L.mth_flatten(self.h, *a, *([None] * 8)):
            raise RuntimeError("flatten failed: " + self.last_error())
        n, m, x = nt.value, nm.value, nx.value
        vertex, normal, uvw = (np.zeros((max(n, 1), 9)) for _ in range(3))
        aabb = np.zeros((max(n, 1), 6))
        material = np.zeros(max(n, 1), dtype=np.int32)
        line_no = np.zeros(max(n, 1), dtype=np.int32)
        mats = np.zeros((max(m, 1), 17))
        whf = np.zeros((max(x, 1), 3), dtype=np.int32)
        assert self.L.mth_flatten(self.h, *a, _ptr(vertex), _ptr(normal), _ptr(uvw), _ptr(aabb),
                                  _ptr(material), _ptr(line_no), _ptr(mats), _ptr(whf))
        textures = []
        for i in range(x):
            w, h, fmt = (int(v) for v in whf[i])
            tex = np.zeros((h, w, 3), dtype=np.uint8 if fmt == MT_TEX_RGB8 else np.float64)
            assert self.L.mth_flatten_texels(self.h, i, _ptr(tex))
            textures.append(dict(texels=tex))
        return dict(tree_depth=t["depth"], node_aabb=t["aabb"], node_center=t["center"],
                    node_first_child=t["first_child"], node_prim_begin=t["prim_begin"],
                    node_prim_count=t["prim_count"], tri_id=t["prim_ids"],
                    tri_vertex=vertex[:n], tri_normal=normal[:n], tri_uvw=uvw[:n],
                    tri_aabb=aabb[:n], tri_material=material[:n], tri_line_no=line_no[:n],
                    materials=[dict(values=mats[i, :16], tex=int(mats[i, 16])) for i in range(m)],
                    textures=textures)

    def get_material(self, name):
        d = np.zeros(16)
        has_tex = C.c_int()
        if not self.L.mth_get_material(self.h, name.encode(), _ptr(d), C.addressof(has_tex)):
            return None
        return d, bool(has_tex.value)

    @property
    def num_materials(self):
        return self.L.mth_num_materials(self.h)

    def render(self, cam, image_w, image_h, chunk=None, debug=False):
        """MythTracer::RayTrace(WorkChunk*)."""
        cx, cy, cw, ch = chunk if chunk else (0, 0, image_w, image_h)
        rgb = np.zeros((max(ch, 0), max(cw, 0), 3), dtype=np.uint8)
        dl = np.zeros((max(ch, 0), max(cw, 0)), dtype=np.int32) if debug else None
        dp = np.zeros((max(ch, 0), max(cw, 0), 3)) if debug else None
        st = np.zeros(8, dtype=np.uint64)
        ms = np.zeros(2)
        cam = _f64(cam)
        ok = self.L.mth_render_chunk(self.h, _ptr(cam), image_w, image_h, cx, cy, cw, ch,
                                     _ptr(rgb), _ptr(dl), _ptr(dp), _ptr(st), _ptr(ms))
        if not ok:
            raise RuntimeError("RayTrace failed: " + self.last_error())
        return dict(rgb=rgb, line=dl, point=dp,
                    counters=dict(zip(STAT_NAMES, (int(x) for x in st))),
                    kernel_ms=float(ms[0]), total_ms=float(ms[1]))

    def render_image(self, cam, image_w, image_h):
        """MythTracer::RayTrace(int, int, Camera*, vector<uint8_t>*)."""
        rgb = np.zeros((image_h, image_w, 3), dtype=np.uint8)
        cam = _f64(cam)
        if not self.L.mth_render_image(self.h, _ptr(cam), image_w, image_h, _ptr(rgb)):
            raise RuntimeError("RayTrace failed: " + self.last_error())
        return rgb

    def frame_loop(self, cam, image_w, image_h, n_frames, dyaw=2.0, collect_stats=False):
        """The frame loop of main_local.cc:51-132 through the facade (host_capi.cc mth_frame_loop): lights pushed again
        every frame, a Camera per frame, RayTrace(W, H, &cam, &bitmap) into ONE vector.  Returns (wall ms per frame,
        the last frame)."""
        rgb = np.zeros((image_h, image_w, 3), dtype=np.uint8)
        ms = np.zeros(n_frames, dtype=np.float64)
        cam = _f64(cam)
        if not self.L.mth_frame_loop(self.h, _ptr(cam), image_w, image_h, n_frames, float(dyaw),
                                     1 if collect_stats else 0, _ptr(ms), _ptr(rgb)):
            raise RuntimeError("RayTrace failed: " + self.last_error())
        return ms, rgb

    def intersect(self, rays):
        rays = _f64(rays).reshape(-1, 6)
        n = rays.shape[0]
        out = dict(tri=np.zeros(n, dtype=np.int32), line=np.zeros(n, dtype=np.int32),
                   t=np.full(n, np.nan), point=np.full((n, 3), np.nan))
        if not self.L.mth_intersect(self.h, n, _ptr(rays), _ptr(out["tri"]), _ptr(out["line"]),
                                    _ptr(out["t"]), _ptr(out["point"])):
            raise RuntimeError("IntersectRays failed: " + self.last_error())
        return out

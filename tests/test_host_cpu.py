"""Host side of the product (C++ facade + C ABI), no GPU needed: OBJ/MTL
parsing, octree construction, sensor set-up, wire formats, library symbols and
error behaviour.  The oracle / golden vectors are the checker."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

import orclib
import quirk_files
from conftest import GOLDEN, ROOT

import mythtracer_amd as M
from mythtracer_amd import binding, scenegen, tiling


@pytest.fixture(scope="module", autouse=True)
def _libs(native_libs):
    return native_libs


def test_c_abi_library_exports_every_declared_symbol():
    """Every function include/mythtracer_hip.h declares is exported."""
    hdr = open(os.path.join(ROOT, "include", "mythtracer_hip.h")).read()
    import re
    declared = set(re.findall(r"\b(mt_[a-z_]+)\s*\(", hdr))
    declared -= {"mt_scene_desc", "mt_scene"}
    assert declared == set(M.HIP_SYMBOLS), declared ^ set(M.HIP_SYMBOLS)
    abi = M.hip_abi()
    for name in M.HIP_SYMBOLS:
        assert getattr(abi.lib, name) is not None
    assert abi.lib.mt_abi_version() == binding.MT_ABI_VERSION


def test_struct_layouts_match_the_header():
    """ctypes mirrors == sizeof() in C (compiled on the fly with gcc)."""
    src = r'''
#include <stdio.h>
#include "mythtracer_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(mt_material), sizeof(mt_light), sizeof(mt_texture),
         sizeof(mt_sensor), sizeof(mt_debug_px), sizeof(mt_stats), sizeof(mt_scene_desc));
  return 0;
}'''
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(td, "s")
        subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), "-o", exe, c])
        sizes = [int(v) for v in subprocess.check_output([exe]).split()]
    want = [ctypes.sizeof(t) for t in (binding.mt_material, binding.mt_light, binding.mt_texture,
                                       binding.mt_sensor)]
    want += [binding.DEBUG_PX_DTYPE.itemsize, ctypes.sizeof(binding.mt_stats),
             ctypes.sizeof(binding.mt_scene_desc)]
    assert sizes == want


def test_no_gpu_means_loud_failure_not_fallback(scenes):
    """Without a usable device every product entry point fails with an error."""
    abi = M.hip_abi()
    if abi.device_count() > 0:
        pytest.skip("a GPU is visible here")
    m = M.MythTracer(scenes["cornell"])
    with pytest.raises(RuntimeError):
        m.prepare()
    with pytest.raises(RuntimeError):
        m.render((50, 50, -120, 0, 0, 0, 60), 8, 8)
    with pytest.raises(RuntimeError):
        m.intersect([[0, 0, 0, 0, 0, 1]])


def test_scene_create_rejects_malformed_descriptions(scenes):
    """Validation happens before any device work, so it is testable on CPU."""
    abi = M.hip_abi()
    m = M.MythTracer(scenes["mini"])
    flat = m.flatten()

    def expect_arg_error(mutate, needle):
        f = dict(flat)
        mutate(f)
        with pytest.raises(RuntimeError) as e:
            abi.scene_create(f)
        assert needle in str(e.value), str(e.value)

    def bad_child(f):
        f["node_first_child"] = f["node_first_child"].copy()
        f["node_first_child"][0] = len(f["node_first_child"]) - 3
    expect_arg_error(bad_child, "first_child")

    def bad_range(f):
        f["node_prim_count"] = f["node_prim_count"].copy()
        f["node_prim_count"][1] += 5
    expect_arg_error(bad_range, "primitive")

    def bad_material(f):
        f["tri_material"] = f["tri_material"].copy()
        f["tri_material"][3] = 99
    expect_arg_error(bad_material, "material index")

    def self_loop(f):
        f["node_first_child"] = f["node_first_child"].copy()
        f["node_first_child"][9] = 1
    expect_arg_error(self_loop, "first_child")

    d = binding.mt_scene_desc()
    d.struct_size = 8
    assert not abi.lib.mt_scene_create(ctypes.byref(d))
    assert "mismatch" in abi.last_error()


@pytest.mark.parametrize("scene", ["cornell", "mini", "mini_nomtl", "room"])
def test_octree_and_triangles_equal_the_oracle(scene, scenes):
    m = M.MythTracer(scenes[scene])
    o = orclib.OracleScene(scenes[scene])
    tm, to = m.tree(), o.tree()
    assert tm["depth"] == to["depth"]
    for k in ("aabb", "first_child", "prim_begin", "prim_count", "prim_ids"):
        assert np.array_equal(tm[k], to[k]), k
    split = tm["first_child"] > 0
    assert np.array_equal(tm["center"][split], to["center"][split])
    dm, lm, hm = m.triangles()
    do, mo, lo = o.triangles()
    assert np.array_equal(dm, do) and np.array_equal(lm, lo)
    assert np.array_equal(hm, (mo >= 0).astype(np.int32))
    assert np.array_equal(m.root_aabb(), o.root_aabb())
    # materials by name
    for name, vals, tex in o.materials():
        got = m.get_material(name)
        assert got is not None and np.array_equal(got[0], vals) and got[1] == (tex >= 0)
    # the flattened streams are the triangles in node-stream order
    f = m.flatten()
    assert np.array_equal(f["tri_vertex"], dm[tm["prim_ids"], 0:9])
    assert np.array_equal(f["tri_aabb"], dm[tm["prim_ids"], 27:33])
    assert np.array_equal(f["tri_line_no"], lm[tm["prim_ids"]])


def test_programmatic_materials_arrive_as_given():
    """The ctypes shims must keep every converted array alive across the call:
    ka, kd, ks and tf are four distinct vectors on the C side, for the facade
    and for the oracle alike (a freed temporary made them all equal once)."""
    import orclib
    ka, kd, ks, tf = (.1, .2, .3), (.4, .5, .6), (.7, .8, .9), (.11, .22, .33)
    m = M.MythTracer()
    m.add_material("a", ka, kd, ks, ns=7, refl=.25, tr=.5, tf=tf, ni=1.5)
    d, has_tex = m.get_material("a")
    assert not has_tex
    assert np.array_equal(d, np.array(ka + kd + ks + (7, .25, .5) + tf + (1.5,)))
    o = orclib.OracleScene()
    o.add_material("a", ka, kd, ks, ns=7, refl=.25, tr=.5, tf=tf, ni=1.5)
    name, od, tex = o.materials()[0]
    assert name == "a" and tex < 0
    assert np.array_equal(od, d)


def _facade_tree_dump(m):
    """The facade's octree in the format of ref_driver's `tree` directive."""
    t = m.tree()
    _, line, _ = m.triangles()
    out = [np.array([len(t["first_child"])], dtype="<i4").tobytes()]
    for i in range(len(t["first_child"])):
        out.append(t["aabb"][i].astype("<f8").tobytes())
        out.append(t["center"][i].astype("<f8").tobytes())
        b, n = int(t["prim_begin"][i]), int(t["prim_count"][i])
        out.append(np.array([1 if t["first_child"][i] != 0 else 0, n], dtype="<i4").tobytes())
        out.append(line[t["prim_ids"][b:b + n]].astype("<i4").tobytes())
    return b"".join(out)


@pytest.mark.parametrize("scene", ["cornell", "mini", "room"])
def test_octree_equals_the_reference_node_for_node(scene, scenes):
    """OctTree::Finalize / Node::AttemptSplit (octtree.cc:16-24,52-135) pinned
    to the COMPILED REFERENCE: ref_driver dumped its finalized tree breadth-first
    (per node: box, centre, child flag, ordered debug_line_no list;
    tests/golden/trees.json + tree_*.npz).  The facade's tree and the oracle's
    must be that tree, byte for byte."""
    import hashlib
    want = json_load("trees.json")[scene]
    m = M.MythTracer(scenes[scene])
    raw = _facade_tree_dump(m)
    assert raw[:4] == np.array([want["n_nodes"]], dtype="<i4").tobytes()
    assert hashlib.sha256(raw).hexdigest() == want["sha256"]
    o = orclib.OracleScene(scenes[scene])
    to, tm = o.tree(), m.tree()
    for k in ("aabb", "center", "first_child", "prim_begin", "prim_count", "prim_ids"):
        assert np.array_equal(to[k], tm[k]), k
    if scene != "room":
        g = np.load(os.path.join(GOLDEN, "tree_%s.npz" % scene), allow_pickle=False)
        tt = m.tree()
        _, line, _ = m.triangles()
        assert np.array_equal(g["aabb"], tt["aabb"]) and np.array_equal(g["center"], tt["center"])
        assert np.array_equal(g["has_children"], (tt["first_child"] != 0).astype(np.int32))
        assert np.array_equal(g["n_prims"], tt["prim_count"])
        assert np.array_equal(g["lines"], line[tt["prim_ids"]])


def json_load(name):
    import json
    return json.load(open(os.path.join(GOLDEN, name)))


def test_wire_bytes_equal_the_reference():
    """WorkChunk::SerializeInput/Output, DeserializeInput's verdicts and
    Camera::Serialize (mythtracer.cc:314-429, camera.cc:71-96) against the bytes
    the compiled reference produced (tests/golden/wire.npz, ref_driver `wire`)."""
    g = np.load(os.path.join(GOLDEN, "wire.npz"), allow_pickle=False)
    L = M.host_lib()
    f6 = np.concatenate([g["image"], g["chunk"]]).astype(np.int32)
    buf = np.zeros(24, dtype=np.uint8)
    assert L.mth_chunk_serialize_input(f6.ctypes.data, buf.ctypes.data) == 24
    assert np.array_equal(buf, g["input_bytes"])
    cam = np.ascontiguousarray(g["cam"], dtype=np.float64)
    cbuf = np.zeros(56, dtype=np.uint8)
    back = np.zeros(7)
    assert L.mth_camera_roundtrip(cam.ctypes.data, cbuf.ctypes.data, back.ctypes.data) == 56
    assert np.array_equal(cbuf, g["camera_bytes"]) and np.array_equal(back, cam)
    got = np.zeros(6, dtype=np.int32)
    for cand, verdict in zip(g["candidates"], g["verdicts"]):
        raw = np.ascontiguousarray(cand.astype("<u4")).view(np.uint8)
        assert L.mth_chunk_deserialize_input(raw.ctypes.data, 24, got.ctypes.data) == int(verdict), list(cand)
    # output packet: the reference rendered this chunk; its bytes = length + the pixels
    ob = g["output_bytes"]
    cw, ch = int(g["chunk"][2]), int(g["chunk"][3])
    rgb = np.ascontiguousarray(ob[4:])
    packet = np.zeros(ob.size, dtype=np.uint8)
    out = np.zeros(rgb.size, dtype=np.uint8)
    n = L.mth_chunk_output_roundtrip(cw, ch, rgb.ctypes.data, rgb.size, packet.ctypes.data, packet.size,
                                     out.ctypes.data)
    assert n == ob.size and np.array_equal(packet, ob) and np.array_equal(out, rgb)
    assert L.mth_chunk_deserialize_output(cw, ch, np.ascontiguousarray(ob).ctypes.data, ob.size) == 1
    # ... and the oracle renders the same pixels for that chunk
    o = orclib.OracleScene(os.path.join(ROOT, "tests", "scenes", "cornell_n.obj"))
    o.set_lights(g["lights"])
    r = o.render(g["cam"], int(g["image"][0]), int(g["image"][1]), chunk=tuple(int(v) for v in g["chunk"]))
    assert np.array_equal(r["rgb"].reshape(-1), rgb)


def test_root_box_always_contains_the_origin():
    """OctTree's root box starts as {0,0,0}-{0,0,0} and only grows (octtree.cc:8-14)."""
    m = M.MythTracer()
    m.add_triangle([[5, 5, 5], [6, 5, 5], [5, 6, 5]])
    assert np.array_equal(m.root_aabb(), [0, 0, 0, 6, 6, 5])
    o = orclib.OracleScene()
    o.add_triangle([[5, 5, 5], [6, 5, 5], [5, 6, 5]])
    assert np.array_equal(o.root_aabb(), m.root_aabb())


def test_split_boundary_and_straddlers():
    """15 triangles stay in the root, 16 split; a straddler stays in the parent;
    coincident triangles all go down together (no depth cap in the reference)."""
    def grid(n):
        tris = []
        for i in range(n):
            x = 1.0 + i
            tris.append([[x, 1, 1], [x + 0.5, 1, 1], [x, 1.5, 1]])
        return tris
    for n, expect_split in ((15, False), (16, True)):
        m, o = M.MythTracer(), orclib.OracleScene()
        for t in grid(n):
            m.add_triangle(t)
            o.add_triangle(t)
        tm, to = m.tree(), o.tree()
        assert (tm["first_child"][0] != 0) == expect_split
        for k in ("first_child", "prim_begin", "prim_count", "prim_ids", "aabb"):
            assert np.array_equal(tm[k], to[k])
    m, o = M.MythTracer(), orclib.OracleScene()
    for t in grid(20) + [[[0.5, 0.5, 0.5], [20, 0.5, 0.5], [0.5, 1.4, 0.9]]] + [[[3, 1, 1], [3.25, 1, 1], [3, 1.25, 1]]] * 18:
        m.add_triangle(t)
        o.add_triangle(t)
    tm, to = m.tree(), o.tree()
    assert tm["depth"] == to["depth"] and tm["depth"] >= 3
    for k in ("first_child", "prim_begin", "prim_count", "prim_ids", "aabb"):
        assert np.array_equal(tm[k], to[k])
    assert 20 in tm["prim_ids"][:tm["prim_count"][0]]  # the straddler stayed at the root


def test_sensor_matches_reference_vectors():
    g = np.load(os.path.join(GOLDEN, "sensor_rays.npz"), allow_pickle=False)
    i = 0
    while "cam%d" % i in g:
        W, H = (int(v) for v in g["size%d" % i])
        cam = g["cam%d" % i]
        assert np.array_equal(binding.sensor(cam, W, H), orclib.sensor(cam, W, H))
        for (x, y), want in zip(g["pix%d" % i], g["dir%d" % i]):
            assert np.array_equal(binding.sensor_ray(cam, W, H, int(x), int(y)), want)
        i += 1


@pytest.mark.parametrize("name", quirk_files.NAMES)
def test_obj_reader_quirks(name, quirk_dir):
    """Same accept/reject decisions as the reference (golden), same triangles as
    the oracle (which is pinned to the reference's renders of these files)."""
    g = np.load(os.path.join(GOLDEN, "obj_quirks.npz"), allow_pickle=False)
    path = os.path.join(quirk_dir, name + ".obj")
    m = M.MythTracer()
    ok = m.load_obj(path)
    assert int(ok) == int(g["ok_" + name][0])
    if ok:
        o = orclib.OracleScene(path)
        dm, lm, hm = m.triangles()
        do, mo, lo = o.triangles()
        assert np.array_equal(dm, do) and np.array_equal(lm, lo)
        assert np.array_equal(hm, (mo >= 0).astype(np.int32))
        for mname, vals, tex in o.materials():
            got = m.get_material(mname)
            assert got is not None and np.array_equal(got[0], vals)


def test_missing_file_and_bounds(tmp_path):
    m = M.MythTracer()
    assert not m.load_obj(str(tmp_path / "nope.obj"))
    p = tmp_path / "oob.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 7 \n")
    assert not M.MythTracer().load_obj(str(p))  # reference: undefined behaviour; here: an error
    assert not orclib.OracleScene().load_obj(str(p))


def test_texture_loaders(tmp_path):
    """PPM / BMP / TGA decode to byte/255.0 texels, rows top to bottom."""
    rgb = (np.arange(2 * 3 * 3) * 13 % 256).astype(np.uint8).reshape(2, 3, 3)
    ppm = tmp_path / "t.ppm"
    ppm.write_bytes(b"P6\n# c\n3 2\n255\n" + rgb.tobytes())
    bmp = tmp_path / "t.bmp"
    row = lambda y: rgb[y, :, ::-1].tobytes() + b"\0" * ((4 - 9 % 4) % 4)
    pix = row(1) + row(0)
    hdr = (b"BM" + (54 + len(pix)).to_bytes(4, "little") + b"\0\0\0\0" + (54).to_bytes(4, "little") +
           (40).to_bytes(4, "little") + (3).to_bytes(4, "little") + (2).to_bytes(4, "little") +
           (1).to_bytes(2, "little") + (24).to_bytes(2, "little") + b"\0" * 24)
    bmp.write_bytes(hdr + pix)
    tga = tmp_path / "t.tga"
    tga.write_bytes(bytes([0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 3, 0, 2, 0, 24, 0x20]) + rgb[:, :, ::-1].tobytes())
    for f in (ppm, bmp, tga):
        (tmp_path / "m.mtl").write_text("newmtl a\nKa 1 1 1\nmap_Ka %s\n" % f.name)
        (tmp_path / "m.obj").write_text("mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nusemtl a\nf 1/1 2/1 3/1 \n")
        m = M.MythTracer(str(tmp_path / "m.obj"))
        fl = m.flatten()
        assert len(fl["textures"]) == 1
        tex = fl["textures"][0]["texels"]
        assert tex.dtype == np.uint8 and np.array_equal(tex, rgb), f.name
    (tmp_path / "m.mtl").write_text("newmtl a\nmap_Ka missing.ppm\n")
    assert not M.MythTracer().load_obj(str(tmp_path / "m.obj"))
    # damaged / unsupported BMPs are refused, not mis-read
    def bmp32(h=2, compression=0, masks=None, data_off=None, hdr_size=40):
        pix = bytes(range(3 * 2 * 4))
        extra = b"".join(m.to_bytes(4, "little") for m in masks) if masks else b""
        off = 54 + len(extra) if data_off is None else data_off
        return (b"BM" + (off + len(pix)).to_bytes(4, "little") + b"\0\0\0\0" + off.to_bytes(4, "little") +
                hdr_size.to_bytes(4, "little") + (3).to_bytes(4, "little") + (h & 0xffffffff).to_bytes(4, "little") +
                (1).to_bytes(2, "little") + (32).to_bytes(2, "little") + compression.to_bytes(4, "little") +
                b"\0" * 20 + extra + pix)
    cases = {"ok32.bmp": (bmp32(), True),
             "bitfields_bgra.bmp": (bmp32(compression=3, masks=(0x00ff0000, 0x0000ff00, 0x000000ff)), True),
             "bitfields_rgb565ish.bmp": (bmp32(compression=3, masks=(0x000000ff, 0x0000ff00, 0x00ff0000)), False),
             "height_int_min.bmp": (bmp32(h=-2 ** 31), False),
             "pixels_inside_header.bmp": (bmp32(data_off=40), False)}
    for name, (blob, ok) in cases.items():
        (tmp_path / name).write_bytes(blob)
        (tmp_path / "m.mtl").write_text("newmtl a\nKa 1 1 1\nmap_Ka %s\n" % name)
        assert M.MythTracer().load_obj(str(tmp_path / "m.obj")) == ok, name


def _png(w, h, ctype, pixels, filters, level, palette=None, idat_split=None, depth=8, interlace=0):
    """A PNG written by hand: per-row filter types as given, zlib level as given
    (0 = stored blocks, 1 = fixed Huffman for small inputs, 9 = dynamic); `pixels` holds one sample per channel
    (values below 2^depth); depth 1 / 2 / 4 packs them, 16 writes them big-endian; interlace = 1 cuts the image
    into the seven sub-images of Adam7."""
    import struct, zlib
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    px = np.asarray(pixels).reshape(h, w, channels).astype(np.int64)
    bpp = max(1, depth * channels // 8)  # the filters' byte distance

    def row_bytes(samples):  # (n, channels) -> packed bytes of one scanline
        flat = samples.reshape(-1)
        if depth == 8:
            return flat.astype(np.uint8)
        if depth == 16:
            return np.stack([flat >> 8, flat & 255], axis=1).reshape(-1).astype(np.uint8)
        per = 8 // depth
        pad = (-len(flat)) % per
        flat = np.concatenate([flat, np.zeros(pad, dtype=np.int64)]).reshape(-1, per)
        out = np.zeros(len(flat), dtype=np.int64)
        for k in range(per):
            out |= flat[:, k] << (8 - depth * (k + 1))
        return out.astype(np.uint8)

    raw = bytearray()
    passes = [(0, 0, 1, 1)] if not interlace else [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]
    row_no = 0
    for (x0, y0, dx, dy) in passes:
        sub = px[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        stride = len(row_bytes(sub[0]))
        prev = np.zeros(stride, dtype=np.int32)
        for y in range(sub.shape[0]):
            cur = row_bytes(sub[y]).astype(np.int32)
            ft = filters[row_no % len(filters)]
            row_no += 1
            left = np.concatenate([np.zeros(bpp, dtype=np.int32), cur[:-bpp]])[:stride]
            upleft = np.concatenate([np.zeros(bpp, dtype=np.int32), prev[:-bpp]])[:stride]
            if ft == 0: pred = 0
            elif ft == 1: pred = left
            elif ft == 2: pred = prev
            elif ft == 3: pred = (left + prev) >> 1
            else:
                p = left + prev - upleft
                pa, pb, pc = abs(p - left), abs(p - prev), abs(p - upleft)
                pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
            raw.append(ft)
            raw += bytes(((cur - pred) & 255).astype(np.uint8))
            prev = cur
    z = zlib.compress(bytes(raw), level)

    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xffffffff)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, interlace))
    if palette is not None:
        out += chunk(b"PLTE", bytes(palette.astype(np.uint8).reshape(-1)))
    parts = [z] if not idat_split else [z[:idat_split], z[idat_split:]]
    for part in parts:
        out += chunk(b"IDAT", part)
    return out + chunk(b"IEND", b"")


def test_png_textures(tmp_path):
    """PNG decoding (own inflate + unfilter): every supported colour type, all
    five scanline filters, stored / fixed / dynamic deflate blocks, IDAT split
    over two chunks; texels = byte / 255.0 of the RGB(A) image, alpha dropped,
    grey replicated, palette looked up -- what SDL's RGBA32 conversion gives
    (texture.cc:88-104).  Damaged files are refused."""
    rnd = np.random.RandomState(7)
    w, h = 13, 9
    pal = rnd.randint(0, 256, size=(17, 3))
    cases = []
    for ctype, ch in ((2, 3), (6, 4), (0, 1), (4, 2), (3, 1)):
        px = rnd.randint(0, 17 if ctype == 3 else 256, size=(h, w, ch))
        # smooth images too: long matches and dynamic Huffman trees
        smooth = ((np.add.outer(np.arange(h), np.arange(w))[:, :, None] * 3 + np.arange(ch)) % (17 if ctype == 3 else 256))
        for name, img, filters, level, split in ((f"t{ctype}_noise_f0", px, [0], 0, None),
                                                 (f"t{ctype}_noise_all", px, [0, 1, 2, 3, 4], 9, 40),
                                                 (f"t{ctype}_smooth_all", smooth, [4, 3, 2, 1, 0], 9, None),
                                                 (f"t{ctype}_smooth_f1", smooth, [1], 1, None)):
            blob = _png(w, h, ctype, img, filters, level, palette=pal if ctype == 3 else None, idat_split=split)
            if ctype in (2, 6):
                rgb = img[:, :, :3]
            elif ctype == 3:
                rgb = pal[img[:, :, 0]]
            else:
                rgb = np.repeat(img[:, :, :1], 3, axis=2)
            cases.append((name + ".png", blob, rgb.astype(np.uint8)))
    (tmp_path / "m.obj").write_text("mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nusemtl a\nf 1/1 2/1 3/1 \n")
    for name, blob, rgb in cases:
        (tmp_path / name).write_bytes(blob)
        (tmp_path / "m.mtl").write_text("newmtl a\nKa 1 1 1\nmap_Ka %s\n" % name)
        m = M.MythTracer(str(tmp_path / "m.obj"))
        tex = m.flatten()["textures"][0]["texels"]
        assert tex.dtype == np.uint8 and np.array_equal(tex, rgb), name
    # Round 4 (SURVEY 8f-3 leftovers): Adam7-interlaced files of every colour type, 16-bit samples (SDL2_image strips them
    # to their high byte, png_set_strip_16), grey and palette below 8 bits (grey scaled to 8 bits: x255 / x85 / x17) --
    # sizes that leave some of the seven passes empty included.  Where PIL reads the file as 8-bit RGB by itself (all
    # but 16-bit grey) its pixels are compared too.
    PIL_Image = pytest.importorskip("PIL.Image")
    more = []
    for (w2, h2) in ((1, 1), (2, 3), (5, 2), (8, 8), (13, 9), (33, 21)):
        for ctype, ch, depths in ((2, 3, (8, 16)), (6, 4, (8, 16)), (0, 1, (1, 2, 4, 8, 16)), (4, 2, (8, 16)), (3, 1, (1, 2, 4, 8))):
            for depth in depths:
                for interlace in (0, 1):
                    if interlace == 0 and depth == 8:
                        continue  # covered above
                    hi = min(1 << depth, 17 if ctype == 3 else 1 << depth)
                    img = rnd.randint(0, hi, size=(h2, w2, ch))
                    blob = _png(w2, h2, ctype, img, [0, 1, 2, 3, 4], 9, palette=pal if ctype == 3 else None, depth=depth, interlace=interlace)
                    s8 = img >> 8 if depth == 16 else img
                    if ctype in (2, 6):
                        rgb = s8[:, :, :3]
                    elif ctype == 3:
                        rgb = pal[img[:, :, 0]]
                    else:
                        g = s8[:, :, :1] * {1: 255, 2: 85, 4: 17, 8: 1, 16: 1}[depth]
                        rgb = np.repeat(g, 3, axis=2)
                    more.append(("r4_%dx%d_t%d_d%d_i%d.png" % (w2, h2, ctype, depth, interlace), blob, rgb.astype(np.uint8), not (ctype in (0, 4) and depth == 16)))
    import io
    for name, blob, rgb, pil_too in more:
        (tmp_path / name).write_bytes(blob)
        (tmp_path / "m.mtl").write_text("newmtl a\nKa 1 1 1\nmap_Ka %s\n" % name)
        m = M.MythTracer(str(tmp_path / "m.obj"))
        tex = m.flatten()["textures"][0]["texels"]
        assert tex.dtype == np.uint8 and np.array_equal(tex, rgb), name
        if pil_too:
            assert np.array_equal(np.array(PIL_Image.open(io.BytesIO(blob)).convert("RGB")), rgb), ("PIL disagrees", name)
    assert len(more) == 6 * (3 + 3 + 9 + 3 + 7)
    good = cases[1][1]
    bad = {"truncated.png": good[:len(good) // 2], "bad_adler.png": good.replace(good[-20:-16], b"\0\0\0\0", 1),
           "interlace2.png": good[:28] + b"\x02" + good[29:], "depth3.png": good[:24] + b"\x03" + good[25:],
           "rgb_depth4.png": _png(4, 4, 2, rnd.randint(0, 16, size=(4, 4, 3)), [0], 9, depth=8)[:24] + b"\x04" + _png(4, 4, 2, rnd.randint(0, 16, size=(4, 4, 3)), [0], 9)[25:]}
    for name, blob in bad.items():
        (tmp_path / name).write_bytes(blob)
        (tmp_path / "m.mtl").write_text("newmtl a\nKa 1 1 1\nmap_Ka %s\n" % name)
        assert not M.MythTracer().load_obj(str(tmp_path / "m.obj")), name

def test_jpeg_textures(tmp_path):
    """Baseline JPEG decoding, pinned bit for bit to libjpeg-turbo through PIL
    (libjpeg defaults = what SDL2_image's libjpeg backend runs behind IMG_Load,
    texture.cc:68): islow IDCT, fancy upsampling, fixed-point YCbCr->RGB.  Grey
    and colour, 4:4:4 / 4:2:2 / 4:2:0, optimised Huffman tables, restart
    markers, sizes that are not a multiple of the MCU, chroma planes narrow
    enough for libjpeg's plain-replication branch.  The reference holds no JPEG
    fixture of its own, so against the reference this stays "parity unpinned"."""
    PIL_Image = pytest.importorskip("PIL.Image")
    rnd = np.random.RandomState(3)
    (tmp_path / "m.obj").write_text("mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nusemtl a\nf 1/1 2/1 3/1 \n")
    (tmp_path / "m.mtl").write_text("newmtl a\nKa 1 1 1\nmap_Ka t.jpg\n")
    f = tmp_path / "t.jpg"
    checked = 0
    for (w, h) in ((1, 1), (3, 3), (4, 5), (6, 2), (13, 9), (16, 16), (33, 17), (64, 48), (1, 17), (17, 1), (200, 120)):
        yy, xx = np.mgrid[0:h, 0:w]
        smooth = np.stack([(xx * 7 + yy * 3) % 256, (xx * 2 + yy * 11) % 256, (xx * yy) % 256], axis=2).astype(np.uint8)
        noise = rnd.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
        for img in (smooth, noise):
            for mode, sub, q, extra in (("RGB", 0, 95, {}), ("RGB", 1, 75, {"optimize": True}), ("RGB", 2, 30, {}),
                                        ("RGB", 2, 100, {"restart_marker_blocks": 3}), ("RGB", 1, 50, {"restart_marker_rows": 1}),
                                        ("L", None, 80, {}), ("L", None, 40, {"optimize": True, "restart_marker_blocks": 2})):
                kw = dict(quality=q, **extra)
                if sub is not None:
                    kw["subsampling"] = sub
                PIL_Image.fromarray(img).convert(mode).save(f, "JPEG", **kw)
                want = np.array(PIL_Image.open(f).convert("RGB"))
                m = M.MythTracer()
                assert m.load_obj(str(tmp_path / "m.obj")), (w, h, mode, sub, q)
                tex = m.flatten()["textures"][0]["texels"]
                assert tex.dtype == np.uint8 and np.array_equal(tex, want), (w, h, mode, sub, q, extra)
                checked += 1
    assert checked == 11 * 2 * 7
    # Round 4 (SURVEY 8f-3 leftovers): PROGRESSIVE files -- spectral selection and successive approximation, DC and AC
    # refinement scans, end-of-band runs, per-component scans -- decode to the same coefficients and therefore to the
    # same pixels as libjpeg-turbo's (jdphuff.c restated); libjpeg's scan script for colour (10 scans) and grey (6),
    # with and without restart intervals and optimised tables
    checked = 0
    for (w, h) in ((1, 1), (3, 3), (6, 2), (13, 9), (16, 16), (33, 17), (64, 48), (200, 120)):
        yy, xx = np.mgrid[0:h, 0:w]
        smooth = np.stack([(xx * 7 + yy * 3) % 256, (xx * 2 + yy * 11) % 256, (xx * yy) % 256], axis=2).astype(np.uint8)
        noise = rnd.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
        for img in (smooth, noise):
            for mode, sub, q, extra in (("RGB", 0, 92, {}), ("RGB", 1, 70, {"optimize": True}), ("RGB", 2, 35, {}),
                                        ("RGB", 2, 98, {"restart_marker_blocks": 3}), ("L", None, 85, {}),
                                        ("L", None, 30, {"restart_marker_rows": 1})):
                kw = dict(quality=q, progressive=True, **extra)
                if sub is not None:
                    kw["subsampling"] = sub
                PIL_Image.fromarray(img).convert(mode).save(f, "JPEG", **kw)
                assert b"\xff\xc2" in f.read_bytes()
                want = np.array(PIL_Image.open(f).convert("RGB"))
                m = M.MythTracer()
                assert m.load_obj(str(tmp_path / "m.obj")), ("progressive", w, h, mode, sub, q)
                tex = m.flatten()["textures"][0]["texels"]
                assert tex.dtype == np.uint8 and np.array_equal(tex, want), ("progressive", w, h, mode, sub, q, extra)
                checked += 1
    assert checked == 8 * 2 * 6
    # truncated headers and other damage are refused, not guessed at
    PIL_Image.fromarray(noise).save(f, "JPEG")
    good = f.read_bytes()
    sof = good.index(b"\xff\xc0")
    for blob in (good[:sof + 6], good[:2], good[:sof + 4] + b"\x0c" + good[sof + 5:],       # cut, bare SOI, 12-bit
                 good[:sof + 9] + b"\x00" + good[sof + 10:],                                # zero components
                 good.replace(b"\xff\xc4", b"\xff\xe5")):                                  # no Huffman tables
        f.write_bytes(blob)
        assert not M.MythTracer().load_obj(str(tmp_path / "m.obj"))


def test_wire_formats():
    """WorkChunk / Camera (de)serialisation, mythtracer.cc:314-429, camera.cc:71-96."""
    L = M.host_lib()
    f6 = np.array([1920, 1080, 128, 256, 128, 128], dtype=np.int32)
    buf = np.zeros(24, dtype=np.uint8)
    assert L.mth_chunk_serialize_input(f6.ctypes.data, buf.ctypes.data) == 24
    assert buf.tobytes() == f6.astype("<u4").tobytes()
    back = np.zeros(6, dtype=np.int32)
    assert L.mth_chunk_deserialize_input(buf.ctypes.data, 24, back.ctypes.data) == 1
    assert np.array_equal(back, f6)
    bad = [[0, 10, 0, 0, 1, 1], [10, 10, 0, 0, 0, 1], [10, 10, 5, 5, 6, 1], [100001, 10, 0, 0, 1, 1],
           [10, 10, 11, 0, 1, 1], [10, 10, 0, 0, 11, 1]]
    for b in bad:
        raw = np.array(b, dtype="<u4").view(np.uint8)
        assert L.mth_chunk_deserialize_input(raw.ctypes.data, 24, back.ctypes.data) == 0, b
    assert L.mth_chunk_deserialize_input(buf.ctypes.data, 23, back.ctypes.data) == 0
    rgb = (np.arange(5 * 3 * 3) % 251).astype(np.uint8)
    packet = np.zeros(4 + rgb.size, dtype=np.uint8)
    out = np.zeros(rgb.size, dtype=np.uint8)
    n = L.mth_chunk_output_roundtrip(5, 3, rgb.ctypes.data, rgb.size, packet.ctypes.data, packet.size,
                                     out.ctypes.data)
    assert n == 4 + rgb.size and np.array_equal(out, rgb)
    assert packet[:4].view("<u4")[0] == rgb.size and np.array_equal(packet[4:], rgb)
    assert L.mth_chunk_deserialize_output(5, 4, packet.ctypes.data, n) == 0   # size mismatch
    assert L.mth_chunk_deserialize_output(5, 3, packet.ctypes.data, 3) == 0   # shorter than the header
    cam = np.array([300.0, 107.0, 40.0, 30.0, 214.0, 0.0, 110.0])
    blob = np.zeros(56, dtype=np.uint8)
    back7 = np.zeros(7)
    assert L.mth_camera_roundtrip(cam.ctypes.data, blob.ctypes.data, back7.ctypes.data) == 56
    assert blob.tobytes() == cam.tobytes() and np.array_equal(back7, cam)


def test_tile_bookkeeping_matches_generatework():
    """tiling.* reproduces main_net_master.cc:195-221 (row-major 128x128
    chunks, clipped at the right/bottom edge) and BlitWorkChunk."""
    W, H, T = 480, 270, 128
    tx, ty = tiling.tile_grid(W, H, T, T)
    assert (tx, ty) == (4, 3)
    rects = [tiling.tile_rect(k, W, H, T, T) for k in range(tx * ty)]
    want = []
    for j in range(0, H, T):
        for i in range(0, W, T):
            want.append((i, j, min(T, W - i), min(T, H - j)))
    assert rects == want
    for world in (1, 2, 3, 5, 8, 16):
        seen = []
        for r in range(world):
            f, s, n = tiling.rank_tiles(W, H, T, T, r, world)
            seen += [f + j * s for j in range(n)]
        assert sorted(seen) == list(range(tx * ty))
    img = np.zeros((H, W, 3), dtype=np.uint8)
    ref = (np.arange(H * W * 3) % 253).astype(np.uint8).reshape(H, W, 3)
    for r in range(3):
        f, s, n = tiling.rank_tiles(W, H, T, T, r, 3)
        slots = np.zeros(n * tiling.slot_bytes(T, T), dtype=np.uint8)
        for j in range(n):
            x, y, cw, ch = tiling.tile_rect(f + j * s, W, H, T, T)
            slots[j * T * T * 3: j * T * T * 3 + cw * ch * 3] = ref[y:y + ch, x:x + cw].reshape(-1)
        tiling.blit_tiles(img, slots, T, T, f, s, n)
    assert np.array_equal(img, ref)


def _compile(tmp_path, src_name, text, extra=()):
    src = tmp_path / src_name
    src.write_text(text)
    exe = tmp_path / (src_name + ".exe")
    inc = os.path.join(ROOT, "mythtracer_amd", "host", "include")
    lib = os.path.join(ROOT, "mythtracer_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", inc, "-I", os.path.join(ROOT, "include"),
                           "-o", str(exe), str(src), "-L", lib, "-lmythtracer_host", "-lmythtracer_hip",
                           "-Wl,-rpath," + lib] + list(extra))
    return exe


def test_math3d_known_answers(tmp_path):
    """The V3D known answers the reference's own unit test checks
    (VerStarting/math3d_test.cc:13-89), as a real asserting program against
    OUR math3d.h."""
    text = r'''
#include <cassert>
#include <cmath>
#include "math3d.h"
using math3d::V3D;
static bool eq(double a, double b) { return std::fabs(a - b) < 1e-7; }
static bool veq(V3D a, V3D b) { return eq(a.v[0], b.v[0]) && eq(a.v[1], b.v[1]) && eq(a.v[2], b.v[2]); }
int main() {
  V3D a{1.0, 2.0, 3.0}, b;
  assert(veq(b, V3D{0, 0, 0}));
  b.x() = 4.0; b.y() = 5.0; b.z() = 6.0;
  V3D c(a); c = b; assert(veq(c, V3D{4, 5, 6}));
  c = a; assert(veq(c += a, V3D{2, 4, 6}));
  c = a; assert(veq(c -= a, V3D{0, 0, 0}));
  c = a; assert(veq(c *= a, V3D{1, 4, 9}));
  c = a; assert(veq(c /= a, V3D{1, 1, 1}));
  c = a; assert(veq(c *= 3.0, V3D{3, 6, 9}));
  c = a;
  assert(veq(c + a, V3D{2, 4, 6}) && veq(c - a, V3D{0, 0, 0}) && veq(c * a, V3D{1, 4, 9}) && veq(c / a, V3D{1, 1, 1}));
  assert(veq(-c, V3D{-1, -2, -3}) && veq(+c, V3D{1, 2, 3}));
  assert(eq((V3D{1, 0, 0}).Length(), 1.0) && eq((V3D{0, 0, 1}).SqrLength(), 1.0));
  c = V3D{1, 2, 3};
  assert(eq(c.Length(), 3.7416573867739413) && eq(c.SqrLength(), 14.0));
  a = V3D{1, 1, 1}; b = V3D{2, 2, 2};
  assert(eq(a.Distance(b), b.Distance(a)) && eq(a.Distance(b), 1.7320508075688772));
  a = V3D{1, 2, 3}; b = V3D{5, 4, 3};
  assert(eq(a.Dot(b), 22.0) && eq(a.Dot(b), b.Dot(a)));
  assert(veq(a.Cross(b), V3D{-6, 12, -6}) && veq(b.Cross(a), V3D{6, -12, 6}));
  b = a; a.Norm();
  assert(veq(a, V3D{0.2672612419124, 0.5345224838248, 0.8017837257372}) && veq(b.DupNorm(), a));
  assert(eq(math3d::Deg2Rad(180.0), M_PI));
  math3d::M4D rx = math3d::M4D::RotationXDeg(90.0);
  V3D r = rx * V3D{0, 1, 0};
  assert(veq(r, V3D{0, 0, 1}));
  return 0;
}'''
    exe = _compile(tmp_path, "m3d.cc", text)
    subprocess.check_call([str(exe)])


@pytest.mark.skipif(not os.path.isdir("/root/reference/VerStarting"), reason="reference tree not present")
def test_reference_drivers_compile_against_our_headers(tmp_path):
    """Drop-in check: the reference's own drivers build, unmodified, against the
    facade headers (syntax + semantic check only; nothing from them is linked or run)."""
    inc = os.path.join(ROOT, "mythtracer_amd", "host", "include")
    for drv in ("main_local.cc",):
        subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-I", inc,
                               "/root/reference/VerStarting/" + drv])


def test_hand_issued_scalar_prefetch_is_safe_on_the_isa():
    """The scan loop issues its box fetches from inline asm; hipcc does not model
    them.  tools/check_asm_prefetch.py compiles the library to ISA and proves that
    nothing touches the in-flight SGPRs before the matching s_waitcnt."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_asm_prefetch.py")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode()

"""Small .obj/.mtl files that exercise the reference parser's behaviours
(VerStarting/objreader.cc), written to tests/scenes/quirks/.  The golden
results (does LoadObj succeed, 24x24 render + first-hit line numbers) come
from the reference itself: tests/golden/make_golden.py -> obj_quirks.npz.

Geometry: a 3x3 grid of points on the plane z = 0, x,y in {-8, 0, 8}; vertex
index = 1 + ix + 3*iy.  The camera looks down +z from (0, 0, -10); the light is
in front of the plane, so shadow rays never meet another triangle (the
reference dereferences a null material for material-less occluders).
"""
from __future__ import annotations

import os

CAM = (0.0, 0.0, -10.0, 0.0, 0.0, 0.0, 90.0)
LIGHTS = [(2.0, 3.0, -6.0, 0.2, 0.2, 0.2, 0.9, 0.9, 0.9, 0.5, 0.5, 0.5)]

GRID = "".join("v %d %d 0\n" % (x, y) for y in (-8, 0, 8) for x in (-8, 0, 8))
NORMAL = "vn 0 0 -1\n"
MTL = """newmtl red
Ka 0.9 0.1 0.1
Kd 0.9 0.1 0.1
Ks 0.2 0.2 0.2
Ns 4
newmtl blue
Ka 0.1 0.1 0.9
Kd 0.1 0.1 0.9
Ks 0.0 0.0 0.0
Ns 1
"""

FILES = {}  # name -> {filename: text}


def case(name, obj, **extra):
    files = {name + ".obj": obj}
    for k, v in extra.items():
        files[k.replace("__", ".")] = v
    FILES[name] = files


case("basic", "mtllib common.mtl\n" + GRID + NORMAL + "usemtl red\nf 1//1 2//1 4//1 \nusemtl blue\nf 5//1 6//1 9//1 8//1 \n",
     common__mtl=MTL)
case("no_trailing_space_tri", GRID + NORMAL + "f 1//1 2//1 4//1 \nf 5//1 6//1 9//1\n")
case("no_trailing_space_quad", GRID + NORMAL + "f 5//1 6//1 9//1 8//1\nf 1//1 2//1 4//1 \n")
case("crlf", (GRID + NORMAL + "f 1//1 2//1 4//1 \nf 5//1 6//1 9//1 8//1 \n").replace("\n", "\r\n"))
case("crlf_eats_space", (GRID + NORMAL + "f 1//1 2//1 4//1\r\n"))
case("hex_octal", GRID + NORMAL + "f 0x1//1 02//01 4//0x1 \nf 05//1 0x6//1 9//1 010//1 \n")
case("forms", GRID + NORMAL + "vt 0 0\nvt 1 0\nvt 0 1\n"
     "f 1/1/1 2/2/1 4/3/1 \nf 2//1 3//1 5//1 \nf 4/1 5/2 7/3 \nf 5 6 8 \nf 6/1/ 9/2/ 8/3/ \n")
case("mixed_normal_indices", GRID + NORMAL + "f 1//1 2 4//1 \nf 5//1 6//1 9//1 8 \n")
case("comments_unknown", "# a comment\n\n   \nfoo 1 2 3\n" + GRID + NORMAL +
     "s 1\ng group name\no object\n#f 1 2 4 \nf 1//1 2//1 4//1 \n  # indented comment\nvx 1 2 3\nf 5//1 6//1 9//1 8//1 \n")
case("leading_space_v", " v 1 2 3\n" + GRID + NORMAL + "f 1//1 2//1 4//1 \n")
case("leading_space_f", GRID + NORMAL + "  f 1//1 2//1 4//1 \nf 5//1 6//1 9//1 8//1 \n")
case("usemtl_unknown", "mtllib common.mtl\n" + GRID + NORMAL +
     "usemtl red\nf 1//1 2//1 4//1 \nusemtl nosuch\nf 5//1 6//1 9//1 8//1 \n", common__mtl=MTL)
case("usemtl_before_mtllib", "usemtl red\nmtllib common.mtl\n" + GRID + NORMAL + "f 1//1 2//1 4//1 \n",
     common__mtl=MTL)
case("long_comment", "# " + "x" * 200 + "\n" + GRID + NORMAL + "f 1//1 2//1 4//1 \nf 5//1 6//1 9//1 8//1 \n")
case("long_vertex_line", "v -8 -8 0" + " " * 140 + "\n" + GRID[len("v -8 -8 0\n"):] + NORMAL +
     "f 1//1 2//1 4//1 \nf 5//1 6//1 9//1 8//1 \n")
case("line_exactly_127", "#" + "y" * 126 + "\n" + GRID + NORMAL + "f 1//1 2//1 4//1 \n")
case("line_exactly_126", "#" + "y" * 125 + "\n" + GRID + NORMAL + "f 1//1 2//1 4//1 \n")
case("mtl_features", "mtllib features.mtl\n" + GRID + NORMAL +
     "usemtl shiny\nf 1//1 2//1 4//1 \nusemtl dup\nf 5//1 6//1 9//1 8//1 \nusemtl plain\nf 2//1 3//1 5//1 \n",
     features__mtl="# material library\nnewmtl shiny\n\tKa 0.3 0.6 0.2\n  Kd 0.3 0.6 0.2\nKs 1 1 1\nNs 30\nd 0.5\nillum 2\n"
                   "Ke 1 1 1\nmap_Kd nothing.png\nNi 1.45\nFoo 12\nnewmtl dup\nKa 1 0 0\nKd 1 0 0\n"
                   "newmtl plain\nKa 0.5 0.5 0.5\nKd 0.25 0.5 0.75\nTf 0.1 0.2 0.3\n"
                   "newmtl dup\nKa 0 1 1\nKd 0 1 1\n")
case("mtllib_missing", "mtllib does_not_exist.mtl\n" + GRID + NORMAL + "f 1//1 2//1 4//1 \n")
case("mtl_without_newmtl", "mtllib bad.mtl\n" + GRID + NORMAL + "f 1//1 2//1 4//1 \n", bad__mtl="Ka 1 1 1\nnewmtl x\n")
case("mtl_indented_newmtl", "mtllib bad2.mtl\n" + GRID + NORMAL + "f 1//1 2//1 4//1 \n", bad2__mtl="  newmtl x\nKa 1 1 1\n")
case("five_vertices", GRID + NORMAL + "f 1//1 2//1 5//1 4//1 7//1 \n")
case("two_vertices", GRID + NORMAL + "f 1//1 2//1 \n")
case("vt_two_components", GRID + NORMAL + "vt 0.25 0.75\nvt 1 0 0.5\nvt 0 1\nf 1/1/1 2/2/1 4/3/1 \n")
case("vt_one_component", GRID + NORMAL + "vt 0.25\nf 1//1 2//1 4//1 \n")
case("v_two_components", "v 1 2\n" + GRID + NORMAL + "f 1//1 2//1 4//1 \n")
case("empty", "")
case("only_comments", "# nothing here\n\n")
case("tabs", (GRID + NORMAL).replace(" ", "\t") + "f\t1//1\t2//1\t4//1\t\nf 5//1\t 6//1 \t9//1 8//1 \t \n")
case("scientific", "v -8e0 -.8E1 0\nv 0.0 -8 0\nv +8 -8 -0\n" + GRID[len("v -8 -8 0\nv 0 -8 0\nv 8 -8 0\n"):] +
     NORMAL + "f 1//1 2//1 4//1 \nf 5//1 6//1 9//1 8//1 \n")
case("long_keyword", "vvvvvvvvvvvvvvvvvvvv 1 2 3\nmtllibbbbbbbbbbbbbbbbb x\n" + GRID + NORMAL + "f 1//1 2//1 4//1 \n")
case("no_normals", GRID + "f 1 2 4 \nf 5 6 9 8 \n")
case("quad_order", GRID + NORMAL + "f 1//1 3//1 9//1 7//1 \n")
case("bad_face_token", GRID + NORMAL + "f 1//1 x 4//1 \n")
case("trailing_garbage_numbers", "v -8 -8 0 1.0 junk\n" + GRID[len("v -8 -8 0\n"):] + NORMAL + "f 1//1 2//1 4//1 \n")

NAMES = sorted(FILES)


def write_all(out_dir: str):
    os.makedirs(out_dir, exist_ok=True)
    for name, files in FILES.items():
        for fn, text in files.items():
            with open(os.path.join(out_dir, fn), "w", newline="") as f:
                f.write(text)
    return out_dir

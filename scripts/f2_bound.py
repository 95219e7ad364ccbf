"""SURVEY 8f-2, measured on the CPU (no GPU needed): what would a PROVABLE early-out of the shadow loop save?
Builds the oracle with -DORC_F2_PROBE into /tmp (oracle/mt_oracle.c, the probe's comment has the rule) and renders
the room frame at 480x270 with the reference's traversal, counting the shadow rays' work below the nodes at which
the outcome "in shadow" is already certain.

  python scripts/f2_bound.py      (about 20 s on 8 cores)"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
so = "/tmp/liboracle_f2probe.so"
subprocess.check_call(["gcc", "-O3", "-std=gnu11", "-fopenmp", "-ffp-contract=off", "-fPIC", "-shared", "-DORC_F2_PROBE",
                       "-o", so, os.path.join(ROOT, "oracle", "mt_oracle.c"), "-lm"])
import orclib
orclib.LIB_PATH, orclib._lib = so, None
from mythtracer_amd import scenegen as sg
info = sg.write_scene("room", "/tmp/mt_scenes")
o = orclib.OracleScene(info["obj"]); o.set_lights(sg.ROOM_LIGHTS)
r = o.render(sg.ROOM_CAMERA, 480, 270)
v = list((ctypes.c_uint64 * 8).in_dll(orclib.lib(), "g_f2"))
print("shadow-loop iterations %d; ending on an opaque hit within the light distance: %d (%.1f %%) -- only those can end early at all" % (v[5], v[6], 100.0 * v[6] / v[5]))
print("their node visits %d, below a deciding node %d (%.2f %%)" % (v[0], v[1], 100.0 * v[1] / v[0]))
print("their triangle tests %d, below a deciding node %d (%.2f %%)" % (v[2], v[3], 100.0 * v[3] / v[2]))

"""How much of the 1080p frame is the pixel column whose rays have a zero direction component (camera on the root's
split plane)?  The same frame with the camera moved off the plane by 1e-3, and turned by one degree; work counters off."""
import os, sys, ctypes, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding
torch.cuda.init(); torch.zeros(1, device="cuda")
W, H = 1920, 1080
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"]); m.set_lights(sg.ROOM_LIGHTS)
buf = torch.zeros(W * H * 3, dtype=torch.uint8, device="cuda")
flat = m.flatten()
for variant in sys.argv[1:] + ["now"]:  # library variants lib/libmythtracer_hip_<name>.so, then the current one
  abi = M.hip_abi() if variant == "now" else M.HipAbi(os.path.join(ROOT, "mythtracer_amd", "lib", "libmythtracer_hip_%s.so" % variant))
  h = abi.scene_create(flat); abi.set_lights(h, sg.ROOM_LIGHTS)
  print("library:", variant)
  for name, cam in   (("on the plane", sg.ROOM_CAMERA), ("off the plane", (200.001,) + tuple(sg.ROOM_CAMERA[1:])), ("yaw 1 degree", sg.ROOM_CAMERA[:4] + (1.0,) + sg.ROOM_CAMERA[5:])):
      sens = binding.sensor(cam, W, H)
      for engine in (1, 2):
          abi.set_engine(h, engine); abi.set_stats(h, False)
          for _ in range(8):
              abi.render_chunk_device(h, sens, W, H, (0, 0, W, H), 5, ctypes.c_void_p(buf.data_ptr()))
          torch.cuda.synchronize(); abi.kernel_times(h)
          for _ in range(24):
              abi.render_chunk_device(h, sens, W, H, (0, 0, W, H), 5, ctypes.c_void_p(buf.data_ptr()))
          torch.cuda.synchronize(); a, b = abi.kernel_times(h); t = a + b
          print("%-14s engine %d: mean %.3f min %.3f max %.3f ms" % (name, engine, t.mean(), t.min(), t.max()), flush=True)

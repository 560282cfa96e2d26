"""Times the host side of mcpt_scene_create (flattening + binned-SAH tree) through mcpt_bvh_dump -- no GPU needed -- and prints a hash of the
tree, so that a change of the builder can be checked for "same tree, faster".  python tools/time_host_build.py chess|chess_high
(MCPT_BVH_VERBOSE=1 prints the builder's own phase times.)"""
import ctypes as C
import hashlib
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np
import mcpt_loader

pkg = mcpt_loader.load()
hb = pkg.hip_backend
name = sys.argv[1] if len(sys.argv) > 1 else "chess"
sd = pkg.scenes.chess_scene(width=64, height=36, spp=1) if name == "chess" else pkg.scenes.chess_high(64, 36, 1)
keep = []
d = hb._make_desc(sd, keep)
L = hb.lib()
i = hb.BvhInfo()
t = time.time()
L.mcpt_bvh_dump(C.byref(d), C.byref(i), None, None, None, None, None)
dt = time.time() - t
n = i.n_nodes
boxes, children, qboxes = np.zeros((n, 12), np.float32), np.zeros((n, 2), np.int32), np.zeros((n, 12), np.uint16)
L.mcpt_bvh_dump(C.byref(d), C.byref(i), boxes.ctypes.data_as(C.c_void_p), children.ctypes.data_as(C.c_void_p), qboxes.ctypes.data_as(C.c_void_p), None, None)
h = hashlib.sha1(boxes.tobytes() + children.tobytes() + qboxes.tobytes()).hexdigest()[:12]
print("%s: %d nodes, height %d, %.1f ms, tree %s" % (name, n, i.stack_entries, dt * 1e3, h))

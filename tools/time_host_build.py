import sys, time, os, hashlib
sys.path.insert(0, os.getcwd())
import numpy as np, mcpt_loader; pkg=mcpt_loader.load()
hb=pkg.hip_backend
import ctypes as C
sd = pkg.scenes.chess_scene(width=64,height=36,spp=1) if sys.argv[1]=='chess' else pkg.scenes.chess_high(64,36,1)
keep=[]; d=hb._make_desc(sd,keep)
L=hb.lib(); i=hb.BvhInfo()
t=time.time(); rc=L.mcpt_bvh_dump(C.byref(d), C.byref(i), None,None,None,None,None); dt=time.time()-t
n=i.n_nodes
boxes, children, qboxes = np.zeros((n, 12), np.float32), np.zeros((n, 2), np.int32), np.zeros((n, 12), np.uint16)
L.mcpt_bvh_dump(C.byref(d), C.byref(i), boxes.ctypes.data_as(C.c_void_p), children.ctypes.data_as(C.c_void_p), qboxes.ctypes.data_as(C.c_void_p), None, None)
h=hashlib.sha1(boxes.tobytes()+children.tobytes()+qboxes.tobytes()).hexdigest()[:12]
print(sys.argv[1],'threads',os.environ.get('MCPT_BUILD_THREADS'),'nodes',n,'height',i.stack_entries,'%.1f ms'%(dt*1e3),'tree',h)

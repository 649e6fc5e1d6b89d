import torch, ctypes as C, sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import harness as H
from terra_amd import api, runtime, scenes
L=runtime.load(); H.build_oracle(); H.set_oracle_math(1)
orc=H.lib("orc")
class Ctr(C.Structure):
    _fields_=[(n,C.c_uint64) for n in ("rays","nodes","box_tests","tri_tests","hits","samples","rand_calls","attr_fetches")]
for integ in (0,1,2,6):
    d=scenes.sponza_hall(48,27,1,integrator=integ)
    orc.fn("orc_counters_reset",None,[])()
    want=H.Unit("orc").render_pixels(d,want_calls=False)
    c=Ctr(); orc.fn("orc_counters_get",None,[C.POINTER(Ctr)])(C.byref(c))
    s=scenes.build_scene(L,d,tree_mode=0); fb=runtime.DeviceFramebuffer(48,27); cam=scenes.camera_of(d)
    runtime.render_device(L,cam,s,fb); torch.cuda.synchronize()
    st=runtime.Stats(); L.get_stats(s,C.byref(st)); st=st.as_dict()
    px=fb.pixels_host()
    print("integ",integ,"equal",np.array_equal(px.view(np.uint32),want["pixels"].view(np.uint32)),"dev mean",px.mean(),"orc mean",want["pixels"].mean(),
          {k:(st[k],getattr(c,k)) for k in ("rays","nodes","tri_tests","hits")})

import sys, ctypes as C, numpy as np
sys.path.insert(0,'/root/repo')
from terra_amd import api, scenes
orc = api.TerraLib(sys.argv[1],'orc_')
f = orc.fn("orc_render_pixels", None, [C.POINTER(api.TerraCamera), C.c_void_p, C.POINTER(api.TerraFramebuffer)] + [C.c_size_t]*4 + [C.c_uint64, C.c_void_p])
for mk, integs in ((scenes.cornell_box,(0,1,2,3,4,5,6)),(scenes.cornell_phong,(0,2)),(scenes.cornell_textured,(1,)),(scenes.cornell_spheres,(0,2))):
    for i in integs:
        d = mk(24,16,2,integrator=i); cam = scenes.camera_of(d); s = scenes.build_scene(orc,d); fb = api.Framebuffer(orc,24,16)
        f(C.byref(cam), s, C.byref(fb.fb), 0,0,24,16, 1, None); fb.destroy(); orc.scene_destroy(s)
# empty and single-triangle scenes
for objs in ([], None):
    d = scenes.SceneDesc(objects=[] if objs is None or objs==[] else objs, width=8, height=8, spp=1)
    s = scenes.build_scene(orc,d); fb = api.Framebuffer(orc,8,8); cam = scenes.camera_of(d)
    f(C.byref(cam), s, C.byref(fb.fb), 0,0,8,8, 1, None); fb.destroy(); orc.scene_destroy(s)
print("asan run done")

import sys, ctypes as C, numpy as np
sys.path.insert(0,'/root/repo')
from terra_amd import api, scenes
lib = api.TerraLib(sys.argv[1],'terra_')
clear = lib.fn("terra_amd_clear_error", None, [])
settree = lib.fn("terra_amd_set_tree_mode", C.c_int, [C.c_void_p, C.c_int])
nodes_fn = lib.fn("terra_amd_scene_bvh_nodes", C.c_int, [C.c_void_p, C.c_void_p, C.c_int])
rs = np.random.RandomState(3)
def soup(n):
    c = rs.uniform(-2,2,size=(n,1,3)); t=(c+rs.uniform(-.5,.5,size=(n,3,3))).astype(np.float32)
    return [scenes.ObjectDesc(t, np.zeros_like(t), np.zeros((n,3,2),np.float32))]
cases = [scenes.cornell_box(8,8,1), scenes.cornell_textured(8,8,1), scenes.cornell_spheres(8,8,1), scenes.sponza_hall(8,8,1)]
for n in (0,1,2,3,5,17,300,4000):
    cases.append(scenes.SceneDesc(objects=soup(n) if n else [], width=8, height=8, spp=1))
def big(d, k):          # every coordinate x k: beyond +-13 units the automatic mode takes the reachability path (inflated boxes, parent links)
    for o in d.objects: o.triangles = (np.asarray(o.triangles, np.float32) * np.float32(k)).astype(np.float32)
    return d
cases += [big(scenes.sponza_hall(8,8,1), 100.0), big(scenes.SceneDesc(objects=soup(600), width=8, height=8, spp=1), 1000.0)]
for d in cases:
    for tm in (0,1,2):
        s = lib.scene_create(); settree(s, tm)
        for od in d.objects:
            o = lib.scene_add_object(s, len(od.triangles)).contents; scenes.fill_object(lib, o, od)
        scenes.apply_options(lib, s, d); lib.scene_commit(s); clear()
        buf = np.zeros((max(1,d.triangle_count)*2, 16), np.float32); nodes_fn(s, buf.ctypes.data, len(buf))
        fb = api.Framebuffer(lib, 8, 8); cam = scenes.camera_of(d)
        lib.render(C.byref(cam), s, C.byref(fb.fb), 0,0,8,8); clear()
        lib.scene_commit(s); lib.scene_clear(s); lib.scene_commit(s); clear()
        fb.destroy(); lib.scene_destroy(s)
print("host asan run done")

import torch, ctypes as C, sys
sys.path.insert(0,'.')
from terra_amd import api, runtime, scenes
lib = runtime.load()
for name, mk in (("cornell", scenes.cornell_box), ("phong", scenes.cornell_phong)):
  for tm in (0, 2, 1):
    d = mk(1920, 1080, 16, bounces=8)
    s = scenes.build_scene(lib, d, tree_mode=tm); fb = runtime.DeviceFramebuffer(d.width, d.height)
    runtime.check(lib.set_sample_split(s, 4))
    runtime.render_device(lib, scenes.camera_of(d), s, fb); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    lib.set_work_counters(s, 0)
    e0.record(); runtime.render_device(lib, scenes.camera_of(d), s, fb); e1.record(); torch.cuda.synchronize()
    st = runtime.Stats(); lib.get_stats(s, C.byref(st)); st = st.as_dict()
    lib.set_work_counters(s, 1); lib.reset_stats(s)
    runtime.render_device(lib, scenes.camera_of(d), s, fb); torch.cuda.synchronize()
    st = runtime.Stats(); lib.get_stats(s, C.byref(st)); st = st.as_dict()
    print(name, "tree mode", tm, "ms", round(e0.elapsed_time(e1), 2), "nodes/ray", round(st["nodes"]/st["rays"], 2), "tri tests/ray", round(st["tri_tests"]/st["rays"], 2), "rays", st["rays"])
    lib.scene_destroy(s)

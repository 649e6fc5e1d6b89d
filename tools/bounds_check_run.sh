cd $GRAFT_REPO_ROOT
export TERRA_AMD_LIB=$GRAFT_REPO_ROOT/terra_amd/libterra_amd_chk.so
timeout -k 10 600 python tools/fuzz_vs_oracle.py 400 41 2>&1 | tail -2
python - <<'PY'
import torch, ctypes as C, sys
sys.path.insert(0,'.')
from terra_amd import api, runtime, scenes
lib = runtime.load(); f = lib.fn("terra_amd_debug_faults", C.c_longlong, [C.c_void_p])
print("library", lib.path.split('/')[-1])
for name, d in (("hall", scenes.sponza_hall(320, 180, 4)), ("spheres", scenes.cornell_spheres(320, 180, 8)), ("cornell", scenes.cornell_box(320, 180, 16))):
    for tm in (0, 1, 2):
        for integ in (0, 1, 2):
            d.integrator = integ
            s = scenes.build_scene(lib, d, tree_mode=tm); fb = runtime.DeviceFramebuffer(d.width, d.height)
            runtime.render_device(lib, scenes.camera_of(d), s, fb); torch.cuda.synchronize()
            print(name, "tree", tm, "integ", integ, "faults", f(s)); lib.scene_destroy(s)
PY
export TERRA_AMD_LIB=$GRAFT_REPO_ROOT/terra_amd/libterra_amd_chkneg.so
python - <<'PY'
import torch, ctypes as C, sys
sys.path.insert(0,'.')
from terra_amd import api, runtime, scenes
lib = runtime.load(); f = lib.fn("terra_amd_debug_faults", C.c_longlong, [C.c_void_p])
d = scenes.sponza_hall(160, 90, 2)
for tm in (0, 1, 2):
    s = scenes.build_scene(lib, d, tree_mode=tm); fb = runtime.DeviceFramebuffer(d.width, d.height)
    runtime.render_device(lib, scenes.camera_of(d), s, fb); torch.cuda.synchronize()
    print("positive control (stack shrunk by 19): tree", tm, "faults", f(s)); lib.scene_destroy(s)
PY
# the fast tree's stack (LDS column + HBM part, trace_device.h fast_push): its rays reach ~10 of the ~28 planned entries, so the control shrinks the plan by 40 -- every push is then out of plan
export TERRA_AMD_LIB=$GRAFT_REPO_ROOT/terra_amd/libterra_amd_chkneg2.so
python - <<'PY'
import torch, ctypes as C, sys
sys.path.insert(0,'.')
from terra_amd import api, runtime, scenes
lib = runtime.load(); f = lib.fn("terra_amd_debug_faults", C.c_longlong, [C.c_void_p])
d = scenes.sponza_hall(160, 90, 2)
for tm in (1, 2):
    s = scenes.build_scene(lib, d, tree_mode=tm); fb = runtime.DeviceFramebuffer(d.width, d.height)
    runtime.render_device(lib, scenes.camera_of(d), s, fb); torch.cuda.synchronize()
    print("positive control (plan shrunk by 40): tree", tm, "faults", f(s)); lib.scene_destroy(s)
PY

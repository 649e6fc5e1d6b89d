"""Quick GPU bring-up check: Cornell through terra_render (host fb) and the device-resident entry,
bit-compared with the oracle in devmath mode. Not a test; see tests/ for the real suite."""
import torch  # first: libterra_amd.so must bind to the HIP runtime torch loaded
import ctypes as C, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from terra_amd import api, scenes

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
amd = api.TerraLib(os.path.join(root, 'terra_amd/libterra_amd.so'))
orc = api.TerraLib(os.path.join(root, 'oracle/liboracle.so'), 'orc_')
last_error = amd.fn('terra_amd_last_error', C.c_char_p, [])
print('devices', amd.fn('terra_amd_device_count', C.c_int, [])())
orc.fn('orc_set_math_mode', None, [C.c_int])(1)
SIG = [C.POINTER(api.TerraCamera), C.c_void_p, C.POINTER(api.TerraFramebuffer)] + [C.c_size_t] * 4 + [C.c_uint64, C.c_void_p]
orp = orc.fn('orc_render_pixels', None, SIG)

def bits(a): return np.ascontiguousarray(a).view(np.uint32)

for name, mk, integ in [('cornell', scenes.cornell_box, 0), ('cornell', scenes.cornell_box, 1), ('cornell', scenes.cornell_box, 2), ('phong', scenes.cornell_phong, 0), ('phong', scenes.cornell_phong, 2)]:
    d = mk(128, 96, 4, integrator=integ)
    cam = scenes.camera_of(d)
    so = scenes.build_scene(orc, d); fo = api.Framebuffer(orc, d.width, d.height)
    orp(C.byref(cam), so, C.byref(fo.fb), 0, 0, d.width, d.height, scenes.FRAME_SEED, None)
    orp(C.byref(cam), so, C.byref(fo.fb), 0, 0, d.width, d.height, scenes.FRAME_SEED, None)
    sa = scenes.build_scene(amd, d); fa = api.Framebuffer(amd, d.width, d.height)
    t = time.time()
    amd.render(C.byref(cam), sa, C.byref(fa.fb), 0, 0, d.width, d.height)
    amd.render(C.byref(cam), sa, C.byref(fa.fb), 0, 0, d.width, d.height)
    dt = time.time() - t
    err = last_error().decode()
    same_acc = np.array_equal(bits(fa.results['acc']), bits(fo.results['acc']))
    same_pix = np.array_equal(bits(fa.pixels), bits(fo.pixels))
    diff = np.abs(fa.pixels.astype(np.float64) - fo.pixels.astype(np.float64))
    nbad = int((bits(fa.results['acc']) != bits(fo.results['acc'])).any(axis=-1).sum())
    print(f'{name} integ {integ}: acc bit-equal {same_acc} pixels bit-equal {same_pix} bad px {nbad} maxdiff {np.nanmax(diff):.3g} mean {fa.pixels.mean():.4f} t {dt:.3f}s err "{err}"')

# throughput probe, device-resident

d = scenes.cornell_box(1920, 1080, 64)
cam = scenes.camera_of(d)
sa = scenes.build_scene(amd, d)
pix = torch.zeros(d.height * d.width * 3, dtype=torch.float32, device='cuda')
res = torch.zeros(d.height * d.width * 4, dtype=torch.int32, device='cuda')
rd = amd.fn('terra_amd_render_device', C.c_int, [C.POINTER(api.TerraCamera), C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_size_t] * 6 + [C.c_void_p, C.c_void_p])
for it in range(3):
    torch.cuda.synchronize(); t = time.time()
    rc = rd(C.byref(cam), sa, pix.data_ptr(), res.data_ptr(), d.width, d.height, 0, 0, d.width, d.height, None, None)
    torch.cuda.synchronize(); dt = time.time() - t
    print('rc', rc, 'render 1080p 64spp: %.3f s  %.1f Msamples/s' % (dt, d.width * d.height * 64 / dt / 1e6), last_error().decode())
class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ['rays','nodes','box_tests','tri_tests','hits','samples','rand_calls','attr_fetches','pixels','launches']]
st = Stats(); amd.fn('terra_amd_get_stats', C.c_int, [C.c_void_p, C.POINTER(Stats)])(sa, C.byref(st))
print({n: getattr(st, n) for n, _ in Stats._fields_})

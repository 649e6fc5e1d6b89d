"""apps/terra_headless.c: OBJ/MTL in, PNG/PPM/PFM/HDR out, Terra.h API only.
The same C source is linked (a) against the compiled reference, here, where a deterministic debug
integrator lets the loader/writers be checked exactly against a direct API render, and (b, gpu)
against libterra_amd.so, where the full Direct render must equal the direct API render bit for bit."""
import struct
import subprocess
import zlib

import numpy as np
import pytest

from terra_amd import api, scenes


def write_obj(d, path, mirror_z=False):
    """exports a SceneDesc; mirror_z writes the right-handed twin (z negated, winding swapped) that the tool's default flip undoes"""
    mtl = path.with_suffix(".mtl")
    with open(path, "w") as f, open(mtl, "w") as m:
        f.write(f"mtllib {mtl.name}\n")
        vi = 1
        for k, o in enumerate(d.objects):
            mt = o.material
            m.write(f"newmtl m{k}\nKd {mt.albedo[0]:.9g} {mt.albedo[1]:.9g} {mt.albedo[2]:.9g}\nKe {mt.emissive[0]:.9g} {mt.emissive[1]:.9g} {mt.emissive[2]:.9g}\n")
            if mt.kind == "phong":
                m.write(f"Ks {mt.specular_color[0]:.9g} {mt.specular_color[1]:.9g} {mt.specular_color[2]:.9g}\nNs {mt.specular_intensity:.9g}\n")
            f.write(f"usemtl m{k}\n")
            for t, n in zip(o.triangles, o.normals):
                order = (0, 2, 1) if mirror_z else (0, 1, 2)
                for c in order:
                    z = -1.0 if mirror_z else 1.0
                    f.write(f"v {t[c][0]:.9g} {t[c][1]:.9g} {t[c][2] * z:.9g}\nvn {n[c][0]:.9g} {n[c][1]:.9g} {n[c][2] * z:.9g}\n")
                f.write(f"f {vi}//{vi} {vi + 1}//{vi + 1} {vi + 2}//{vi + 2}\n")
                vi += 3


def write_obj_quads(d, path):
    """same scene, written the way exporters do: shared vertex pool per object, quad faces (fan-triangulated by the
    loader), relative (negative) indices, v/vt/vn triples, comments and blank lines"""
    mtl = path.with_suffix(".mtl")
    with open(path, "w") as f, open(mtl, "w") as m:
        f.write(f"# quads + relative indices\n\nmtllib {mtl.name}\n")
        for k, o in enumerate(d.objects):
            mt = o.material
            m.write(f"newmtl m{k}\nKd {mt.albedo[0]:.9g} {mt.albedo[1]:.9g} {mt.albedo[2]:.9g}\nKe {mt.emissive[0]:.9g} {mt.emissive[1]:.9g} {mt.emissive[2]:.9g}\n")
            f.write(f"o obj{k}\nusemtl m{k}\n")
            t, n = o.triangles, o.normals
            assert len(t) % 2 == 0
            for q in range(0, len(t), 2):
                assert np.array_equal(t[q][0], t[q + 1][0]) and np.array_equal(t[q][2], t[q + 1][1])      # (a,b,c),(a,c,d)
                corners = [(t[q][0], n[q][0]), (t[q][1], n[q][1]), (t[q][2], n[q][2]), (t[q + 1][2], n[q + 1][2])]
                for p, nn in corners:
                    f.write(f"v {p[0]:.9g} {p[1]:.9g} {p[2]:.9g}\nvt 0 0\nvn {nn[0]:.9g} {nn[1]:.9g} {nn[2]:.9g}\n")
                f.write("f -4/-4/-4 -3/-3/-3 -2/-2/-2 -1/-1/-1\n")


def read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = map(int, f.readline().split())
        assert float(f.readline()) < 0
        return np.frombuffer(f.read(), np.float32).reshape(h, w, 3)[::-1]


def read_png(path):
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    o, idat, w, h = 8, b"", 0, 0
    while o < len(b):
        n, typ = struct.unpack(">I4s", b[o:o + 8]); data = b[o + 8:o + 8 + n]
        assert struct.unpack(">I", b[o + 8 + n:o + 12 + n])[0] == zlib.crc32(typ + data)
        if typ == b"IHDR":
            w, h, depth, ctype = struct.unpack(">IIBB", data[:10]); assert (depth, ctype) == (8, 2)
        elif typ == b"IDAT":
            idat += data
        o += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 3 * w)
    assert not raw[:, 0].any()
    return raw[:, 1:].reshape(h, w, 3)


def build_tool(H, tmp_path, against):
    exe = tmp_path / f"terra_headless_{against}"
    if against == "ref":
        cmd = ["gcc", "-std=gnu11", "-w", "-O1", "-I/root/reference/include", "-I/root/reference/src", str(H.ROOT / "apps/terra_headless.c"),
               str(H.REF_SO), f"-Wl,-rpath,{H.REF_SO.parent}", "-lm", "-o", str(exe)]
    else:
        cmd = ["gcc", "-std=gnu11", "-Wall", "-O1", f"-I{H.ROOT / 'include'}", str(H.ROOT / "apps/terra_headless.c"), f"-L{H.ROOT / 'terra_amd'}", "-lterra_amd",
               f"-Wl,-rpath,{H.ROOT / 'terra_amd'}", "-lm", "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_tool_against_the_compiled_reference(H, ref_lib, tmp_path):
    exe = build_tool(H, tmp_path, "ref")
    d = scenes.cornell_phong(80, 60, 1, integrator=api.kTerraIntegratorDebugNormals, jitter=0.0, tonemap=api.kTerraTonemappingOperatorNone)
    want = H.Unit("ref").render_pixels(d, want_calls=False)["pixels"]
    for mirror, extra in ((False, ["--no-flip-z"]), (True, [])):
        obj = tmp_path / f"cornell_{int(mirror)}.obj"
        write_obj(d, obj, mirror_z=mirror)
        out = tmp_path / f"n_{int(mirror)}.pfm"
        r = subprocess.run([str(exe), str(obj), str(out), "--width", "80", "--height", "60", "--spp", "1", "--integrator", "normals", "--tonemap", "none", "--normals", "file"] + extra, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr + r.stdout
        assert "32 triangles, 6 materials" in r.stdout
        assert np.array_equal(read_pfm(out), want), mirror
    # the other writers carry the same pixels (PNG/PPM: clamp x 255)
    for ext in ("png", "ppm", "hdr"):
        out = tmp_path / f"img.{ext}"
        r = subprocess.run([str(exe), str(obj), str(out), "--width", "80", "--height", "60", "--spp", "1", "--integrator", "normals", "--tonemap", "none", "--normals", "file"], capture_output=True, text=True)
        assert r.returncode == 0 and out.stat().st_size > 1000
    bytes_want = (np.clip(want, 0, 1) * np.float32(255)).astype(np.uint8)
    assert np.array_equal(read_png(tmp_path / "img.png"), bytes_want)
    ppm = open(tmp_path / "img.ppm", "rb").read()
    assert ppm.startswith(b"P6\n80 60\n255\n") and np.array_equal(np.frombuffer(ppm[len(b"P6\n80 60\n255\n"):], np.uint8).reshape(60, 80, 3), bytes_want)


def test_tool_quads_relative_indices_and_missing_files(H, ref_lib, tmp_path):
    exe = build_tool(H, tmp_path, "ref")
    d = scenes.cornell_box(64, 48, 1, integrator=api.kTerraIntegratorDebugDepth, jitter=0.0)
    want = H.Unit("ref").render_pixels(d, want_calls=False)["pixels"]
    obj = tmp_path / "quads.obj"; write_obj_quads(d, obj)
    out = tmp_path / "q.pfm"
    r = subprocess.run([str(exe), str(obj), str(out), "--width", "64", "--height", "48", "--spp", "1", "--integrator", "depth", "--tonemap", "none", "--no-flip-z", "--normals", "file"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "32 triangles" in r.stdout and np.array_equal(read_pfm(out), want)
    # errors are reported, not crashed on
    r = subprocess.run([str(exe), str(tmp_path / "missing.obj"), str(out)], capture_output=True, text=True)
    assert r.returncode != 0 and "cannot open" in r.stderr
    (tmp_path / "empty.obj").write_text("# nothing\n")
    r = subprocess.run([str(exe), str(tmp_path / "empty.obj"), str(out)], capture_output=True, text=True)
    assert r.returncode != 0
    # an unknown extension gets PNG, as the reference's exporter assumes (satellite/src/Visualization.cpp:313-316)
    r = subprocess.run([str(exe), str(obj), str(tmp_path / "x.bmp"), "--width", "32", "--height", "24", "--spp", "1", "--integrator", "normals"], capture_output=True, text=True)
    assert r.returncode == 0 and open(tmp_path / "x.bmp", "rb").read(8) == b"\x89PNG\r\n\x1a\n"


@pytest.mark.gpu
def test_tool_against_the_product(H, amd_lib, orc_lib, devmath_mode, tmp_path):
    """OBJ/MTL -> tool (linked against libterra_amd.so) -> PFM == the ORACLE's render of the same scene description, bit for bit (device-twin math: the parity
    leg of SURVEY 8f N1 on the GPU; reference caller satellite/src/Scene.cpp:133-245), for both tree modes -- and == a direct API render through the product"""
    from terra_amd import runtime
    from test_gpu_render import render_host
    L = runtime.load()
    exe = build_tool(H, tmp_path, "amd")
    d = scenes.cornell_phong(96, 64, 4, integrator=api.kTerraIntegratorDirect, jitter=0.5, tonemap=api.kTerraTonemappingOperatorReinhard, environment=(0.4, 0.52, 1.0))
    want = H.Unit("orc").render_pixels(d, want_calls=False)["pixels"]          # the checker: the oracle, not the product
    assert np.array_equal(render_host(L, d)["pixels"].view(np.uint32), want.view(np.uint32))
    obj = tmp_path / "cornell.obj"
    write_obj(d, obj, mirror_z=True)
    out = tmp_path / "direct.pfm"
    args = [str(exe), str(obj), str(out), "--width", "96", "--height", "64", "--spp", "4", "--bounces", "8", "--integrator", "direct", "--tonemap", "reinhard", "--jitter", "0.5", "--tile", "32", "--normals", "file"]
    r = subprocess.run(args, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    assert np.array_equal(read_pfm(out).view(np.uint32), want.view(np.uint32))
    r = subprocess.run(args + ["--fast-tree"], capture_output=True, text=True)
    assert r.returncode == 0 and np.array_equal(read_pfm(out).view(np.uint32), want.view(np.uint32))
    # --gpus N: the in-process multi-device path (scene replicas, tiles dealt to the devices, one RCCL gather issued from C); one device here, the same calls
    r = subprocess.run(args + ["--gpus", "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    assert np.array_equal(read_pfm(out).view(np.uint32), want.view(np.uint32))
    r = subprocess.run(args + ["--gpus", "64"], capture_output=True, text=True)
    assert r.returncode == 69 and "visible" in r.stderr


# ---------------------------------------------------------------------------
# the reference importer's policy (Apollo.h under Scene.cpp's options), pinned by HAND-DERIVED fixtures: Apollo.h does not compile
# here, so every expected number below was worked out on paper from the rules in apps/terra_headless.c's header comment
# ---------------------------------------------------------------------------

def import_dump(H, tmp_path, name, obj_text, mtl_text=None, extra=()):
    from terra_amd import build
    build.build()
    exe = build_tool(H, tmp_path, "amd")
    (tmp_path / f"{name}.obj").write_text(obj_text)
    if mtl_text is not None:
        (tmp_path / f"{name}.mtl").write_text(mtl_text)
    dump = tmp_path / f"{name}.dump"
    r = subprocess.run([str(exe), str(tmp_path / f"{name}.obj"), str(tmp_path / "unused.png"), "--dump-scene", str(dump), "--no-render"] + list(extra), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    objs = []
    for line in open(dump):
        w = line.split()
        if w[0] == "object":
            objs.append(dict(attributes=int(w[5]), tris=[]))
        elif w[0] == "m":      # m <preset> <attribute values, 3 per slot> e <emissive> ior <ior>
            e = w.index("e")
            objs[-1].update(preset=w[1], values=np.array(w[2:e], np.float32).reshape(-1, 3), emissive=np.array(w[e + 1:e + 4], np.float32), ior=np.float32(w[e + 5]))
        else:
            v = np.array(w[1:10] + w[11:20] + w[21:27], np.float32)
            objs[-1]["tris"].append(dict(p=v[0:9].reshape(3, 3), n=v[9:18].reshape(3, 3), uv=v[18:24].reshape(3, 2)))
    return objs


R = np.float32(1.0) / np.sqrt(np.float32(2.0), dtype=np.float32)       # 0.70710677: 1 / |(1, 0, -1)| in float


def renorm(v):
    """the smooth path normalises the SUM of the adjacent face normals, in float: a vertex with one adjacent face gets that face's
    (already unit) normal divided by its own float length once more (Apollo.h:1527-1533)"""
    f = np.float32
    x, y, z = f(v[0]), f(v[1]), f(v[2])
    ln = np.sqrt(f(f(f(x * x) + f(y * y)) + f(z * z)), dtype=np.float32)
    return np.array([x / ln, y / ln, z / ln], np.float32)


def test_import_flat_triangle_flip_and_winding(H, amd_lib, tmp_path):
    # right-handed CCW triangle facing +z; no `s`: face normal per vertex; `vn` in the file is ignored
    o = import_dump(H, tmp_path, "tri", "vn 1 0 0\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1//1 2//1 3//1\n")
    assert len(o) == 1 and len(o[0]["tris"]) == 1 and o[0]["attributes"] == 1          # no material: diffuse with one attribute
    t = o[0]["tris"][0]
    assert np.array_equal(t["p"], [[0, 1, 0], [1, 0, 0], [0, 0, 0]])                     # z negated (all zero here), (a, b, c) -> (c, b, a)
    assert np.array_equal(t["n"], [[0, 0, -1]] * 3)                                       # cross((1,-1,0), (0,-1,0)) = (0, 0, -1): faces -z after the flip
    assert not t["uv"].any()
    # without the flip the triangle stays as written and faces +z
    o = import_dump(H, tmp_path, "tri2", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n", extra=["--no-flip-z"])
    assert np.array_equal(o[0]["tris"][0]["p"], [[0, 0, 0], [1, 0, 0], [0, 1, 0]]) and np.array_equal(o[0]["tris"][0]["n"], [[0, 0, 1]] * 3)


def test_import_smooth_group_shares_positions(H, amd_lib, tmp_path):
    # two triangles at a right angle sharing the edge (0,0,0)-(0,1,0), `s 1`: the shared corners get normalize(nA + nB)
    text = "s 1\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\nf 1 2 3\nf 1 3 4\n"
    o = import_dump(H, tmp_path, "smooth", text)
    a, b = o[0]["tris"]
    assert np.array_equal(a["p"], [[0, 1, 0], [1, 0, 0], [0, 0, 0]]) and np.array_equal(b["p"], [[0, 0, -1], [0, 1, 0], [0, 0, 0]])
    shared = renorm([1, 0, -1])                                                           # nA = (0,0,-1) + nB = (1,0,0), normalised
    assert np.array_equal(a["n"], [shared, [0, 0, -1], shared])                           # (0,1,0) shared, (1,0,0) own, (0,0,0) shared
    assert np.array_equal(b["n"], [[1, 0, 0], shared, shared])
    # the same file with `s off`: face normals, nothing shared
    o = import_dump(H, tmp_path, "flat", text.replace("s 1", "s off"))
    assert np.array_equal(o[0]["tris"][0]["n"], [[0, 0, -1]] * 3) and np.array_equal(o[0]["tris"][1]["n"], [[1, 0, 0]] * 3)
    # `s 2` is not smooth for Apollo (only a token starting with '1' is), `s 10` is
    o = import_dump(H, tmp_path, "s2", text.replace("s 1", "s 2"))
    assert np.array_equal(o[0]["tris"][1]["n"], [[1, 0, 0]] * 3)
    o = import_dump(H, tmp_path, "s10", text.replace("s 1", "s 10"))
    assert np.array_equal(o[0]["tris"][1]["n"], [[1, 0, 0], shared, shared])


def test_import_quad_adjacency_is_per_parsed_corner(H, amd_lib, tmp_path):
    # a non-planar smooth quad p0 p1 p2 p3 = fan (p0,p1,p2), (p0,p2,p3): Apollo records, for every corner, only the triangle being
    # formed while that corner is parsed -- p0, p1, p2 count for triangle 0 only, p3 for triangle 1 only
    o = import_dump(H, tmp_path, "quad", "s 1\nv 0 0 0\nv 1 0 0\nv 1 1 1\nv 0 1 0\nf 1 2 3 4\n")
    t0, t1 = o[0]["tris"]
    assert np.array_equal(t0["p"], [[1, 1, -1], [1, 0, 0], [0, 0, 0]]) and np.array_equal(t1["p"], [[0, 1, 0], [1, 1, -1], [0, 0, 0]])
    n0 = renorm([0, -R, -R])          # face normal cross((0,-1,1), (-1,-1,1)) / sqrt(2) = (0,-R,-R), then the smooth path's own normalisation
    n1 = renorm([-R, 0, -R])          # cross((1,0,-1), (0,-1,0)) / sqrt(2) = (-R,0,-R)
    assert np.array_equal(t0["n"], [n0, n0, n0])
    assert np.array_equal(t1["n"], [n1, n0, n0])    # p3 -> triangle 1; p2 and p0 keep triangle 0's normal (a conventional smooth normal would mix both)


def test_import_groups_and_materials(H, amd_lib, tmp_path):
    mtl = "newmtl red\nKd 1 0 0\nnewmtl shiny\nKd 0.5 0.5 0.5\nKs 0.5 0.5 0.5\nNs 20\nnewmtl lamp\nKd 0 0 0\nKe 5 5 5\nillum specular\n"
    obj = ("mtllib groups.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\n"
           "g first\nusemtl red\nf 1 2 3\nusemtl shiny\nf 1 3 4\n"         # one group: ONE object, its material is the last usemtl (Apollo.h:1101-1130)
           "o second\nusemtl lamp\nvt 0.25 0.75\nf 1/1 2/1 4/1\n"
           "g empty\n")                                                        # a group without faces makes no object
    o = import_dump(H, tmp_path, "groups", obj, mtl)
    assert [len(x["tris"]) for x in o] == [2, 1]
    # shiny has Ks > 0 but no Apollo `illum` word: its class stays invalid and Scene.cpp's switch falls through to diffuse; lamp: `illum specular` -> Phong
    assert (o[0]["preset"], o[0]["attributes"]) == ("diffuse", 1) and (o[1]["preset"], o[1]["attributes"]) == ("phong", 4)
    assert np.array_equal(o[0]["values"], [[0.5, 0.5, 0.5]]) and np.array_equal(o[1]["emissive"], [5, 5, 5])
    assert np.array_equal(o[1]["tris"][0]["uv"], [[0.25, 0.75]] * 3)
    # this repo's other policy groups by material instead, keeps vn and reads Ks > 0 as Phong
    o = import_dump(H, tmp_path, "groups_file", obj, mtl.replace("groups", "groups_file"), extra=["--normals", "file"])
    assert sorted(len(x["tris"]) for x in o) == [1, 1, 1]
    assert sorted(x["preset"] for x in o) == ["diffuse", "phong", "phong"]


MTL_CLASSES = """newmtl a_diffuse
Kd 0.1 0.2 0.3
illum diffuse
newmtl b_specular
Kd 0.4 0.5 0.6
Ks 0.7 0.8 0.9
Ns 32
illum specular
newmtl c_mirror
Kd 0.11 0.12 0.13
Ks 1 1 1
illum mirror
newmtl d_pbr
Kd 0.21 0.22 0.23
Pr 0.5
Pm 1
illum pbr
newmtl e_disney
Kd 0.31 0.32 0.33
illum disney
newmtl f_standard_mtl
Kd 0.41 0.42 0.43
Ks 0.5 0.5 0.5
Ns 100
illum 2
newmtl g_lamp
Kd 0 0 0
Ke 7 8 9
illum diffuse
"""


def test_import_material_class_table(H, amd_lib, tmp_path):
    """one object per Apollo BSDF class (satellite/include/Apollo.h:80-87, set only by `illum <word>`, :877-897) through the reference client's
    switch (satellite/src/Scene.cpp:193-230): specular -> Phong with (specular colour, albedo, exponent x3, pick 0) in TerraPresets.h's slot
    order; diffuse, mirror, pbr, disney and the invalid class (standard MTL's `illum 2`) -> diffuse with Kd; Ke -> emissive; ior 1.5 always.
    HAND-DERIVED from those lines (restated, not executed)."""
    names = ["a_diffuse", "b_specular", "c_mirror", "d_pbr", "e_disney", "f_standard_mtl", "g_lamp"]
    obj = "mtllib classes.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\n" + "".join(f"g {n}\nusemtl {n}\nf 1 2 3\n" for n in names)
    from terra_amd import build
    o = import_dump(H, tmp_path, "classes", obj, MTL_CLASSES)
    assert len(o) == 7 and all(float(x["ior"]) == 1.5 for x in o)
    f = np.float32
    want = {0: [[0.1, 0.2, 0.3]], 2: [[0.11, 0.12, 0.13]], 3: [[0.21, 0.22, 0.23]], 4: [[0.31, 0.32, 0.33]], 5: [[0.41, 0.42, 0.43]], 6: [[0, 0, 0]]}
    for k, kd in want.items():
        assert o[k]["preset"] == "diffuse" and o[k]["attributes"] == 1 and np.array_equal(o[k]["values"], np.array(kd, f)), names[k]
    # Phong slots (include/TerraPresets.h): 0 specular colour, 1 albedo, 2 exponent (to_constant(float) -> all three), 3 the sample-pick scratch = 0
    assert o[1]["preset"] == "phong" and np.array_equal(o[1]["values"], np.array([[0.7, 0.8, 0.9], [0.4, 0.5, 0.6], [32, 32, 32], [0, 0, 0]], f))
    assert np.array_equal(o[6]["emissive"], [7, 8, 9]) and all(not o[k]["emissive"].any() for k in range(6))
    # the reference warns for the classes it cannot shade and goes on (Scene.cpp:215-220); so does the tool
    exe = build_tool(H, tmp_path, "amd")
    r = subprocess.run([str(exe), str(tmp_path / "classes.obj"), str(tmp_path / "unused.png"), "--no-render"], capture_output=True, text=True)
    assert r.returncode == 0
    for word, name in (("mirror", "c_mirror"), ("pbr", "d_pbr"), ("disney", "e_disney"), ("unclassified", "f_standard_mtl")):
        assert any(word in ln and name in ln and "Defaulting to diffuse" in ln for ln in r.stderr.splitlines()), (word, r.stderr)
    assert "a_diffuse" not in r.stderr and "b_specular" not in r.stderr
    # this repo's own policy for standard MTL files reads Ks > 0 as Phong when there is no Apollo word
    o2 = import_dump(H, tmp_path, "classes_file", obj.replace("classes.mtl", "classes_file.mtl"), MTL_CLASSES, extra=["--normals", "file"])
    by_kd = {tuple(round(float(v), 2) for v in x["values"][1 if x["preset"] == "phong" else 0]): x["preset"] for x in o2}
    assert by_kd[(0.41, 0.42, 0.43)] == "phong" and by_kd[(0.11, 0.12, 0.13)] == "diffuse" and by_kd[(0.4, 0.5, 0.6)] == "phong"


def test_import_negative_indices_and_multiple_usemtl(H, amd_lib, tmp_path):
    """relative (negative) indices count back from the elements read so far (the OBJ rule; -1 = the last `v` / `vt` before the face);
    several `usemtl` inside one group leave ONE object whose material is the last of them (Apollo.h:1101-1130)"""
    mtl = "newmtl one\nKd 1 0 0\nillum diffuse\nnewmtl two\nKd 0 1 0\nillum diffuse\nnewmtl three\nKd 0 0 1\nillum specular\n"
    obj = ("mtllib neg.mtl\ng all\nusemtl one\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0.5 0.25\nf -3/-1 -2/-1 -1/-1\n"
           "usemtl two\nv 0 0 2\nf -4 -3 -1\nusemtl three\nf 1 2 -1\n")
    o = import_dump(H, tmp_path, "neg", obj, mtl)
    assert len(o) == 1 and len(o[0]["tris"]) == 3 and o[0]["preset"] == "phong" and np.array_equal(o[0]["values"][1], [0, 0, 1])
    t0, t1, t2 = o[0]["tris"]
    assert np.array_equal(t0["p"], [[0, 1, 0], [1, 0, 0], [0, 0, 0]]) and np.array_equal(t0["uv"], [[0.5, 0.25]] * 3)      # (c, b, a), z negated
    assert np.array_equal(t1["p"], [[0, 0, -2], [1, 0, 0], [0, 0, 0]])                                                      # -4 -3 -1 = v1 v2 v4
    assert np.array_equal(t2["p"], [[0, 0, -2], [1, 0, 0], [0, 0, 0]]) and not t1["uv"].any()                               # 1 2 -1  = v1 v2 v4


def test_help_states_that_the_importer_is_restated(H, amd_lib, tmp_path):
    exe = build_tool(H, tmp_path, "amd")
    r = subprocess.run([str(exe), "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "restated, not executed" in r.stdout and "Apollo.h" in r.stdout

"""The drop-in boundary: every entry point declared in include/*.h is exported by
terra_amd/libterra_amd.so with C linkage, the headers compile as C and C++, and
the struct layouts are the reference's. No compute calls."""
import ctypes as C
import re
import subprocess

import pytest

from terra_amd import api

DECL = re.compile(r"\b(terra_[a-z0-9_]+)\s*\(")


def declared_functions(H):
    names = set()
    for header in ("Terra.h", "TerraPresets.h", "terra_amd.h"):
        text = (H.ROOT / "include" / header).read_text()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"//[^\n]*", "", text)
        text = "\n".join(l for l in text.splitlines() if not l.lstrip().startswith("#"))
        text = text.replace('extern "C" {', "").replace("}", ";")
        for stmt in text.split(";"):
            s = " ".join(stmt.split())
            if not s or s.startswith("#") or "typedef" in s or "static" in s or "TERRA_ABI_ASSERT" in s or "{" in s:
                continue
            m = DECL.search(s)
            if m and "(" in s and s.rstrip().endswith(")"):
                names.add(m.group(1))
    return sorted(names)


def test_every_declared_symbol_is_exported(H, amd_lib):
    names = declared_functions(H)
    assert len(names) >= 27 + 25, names        # Terra.h's 25 + 2 preset inits + the terra_amd_* extension
    for required in ("terra_render", "terra_scene_commit", "terra_bsdf_phong_init", "terra_amd_render_device", "terra_amd_unit_watertight", "terra_amd_pack_tiles"):
        assert required in names
    missing = [n for n in names if not amd_lib.has(n)]
    assert not missing, missing


def test_preset_marker_symbols_exported(amd_lib):
    # non-static in the reference too (src/TerraPresets.c:34,47,52,84,108,125)
    for n in ("terra_bsdf_diffuse_sample", "terra_bsdf_diffuse_pdf", "terra_bsdf_diffuse_eval", "terra_bsdf_phong_sample", "terra_bsdf_phong_pdf", "terra_bsdf_phong_eval"):
        assert amd_lib.has(n)


def test_library_does_not_depend_on_the_oracle(H):
    out = subprocess.run(["ldd", str(H.AMD_SO)], capture_output=True, text=True).stdout
    assert "oracle" not in out and "terra_ref" not in out
    syms = subprocess.run(["nm", "-D", str(H.AMD_SO)], capture_output=True, text=True).stdout
    assert " orc_" not in syms and " ref_" not in syms


@pytest.mark.parametrize("compiler,std", [("gcc", "-std=c11"), ("g++", "-std=c++17")])
def test_headers_compile_and_layout_asserts_hold(H, tmp_path, compiler, std):
    src = tmp_path / ("t.c" if compiler == "gcc" else "t.cpp")
    src.write_text('#include "Terra.h"\n#include "TerraPresets.h"\n#include "terra_amd.h"\nint main(void){TerraFloat3 a=terra_f3_set(1,2,3);TerraFloat3 b=terra_normf3(&a);return b.x>2;}\n')
    r = subprocess.run([compiler, std, "-Wall", "-Werror", "-fsyntax-only", f"-I{H.ROOT / 'include'}", str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_ctypes_mirror_sizes():
    for t, size in api.ABI_SIZES.items():
        assert C.sizeof(t) == size
    assert api.TerraShadingSurface.normal.offset == 64 and api.TerraShadingSurface.attributes.offset == 92
    assert api.TerraMaterial.attributes.offset == 72 and api.TerraMaterial.attributes_count.offset == 392
    assert api.TerraObject.material.offset == 24
    assert api.TerraSceneOptions.samples_per_pixel.offset == 64 and api.TerraSceneOptions.gamma.offset == 92

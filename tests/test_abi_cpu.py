"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/fovpt.h declares, layouts match, the C++ shim compiles, and the product fails loudly
(no CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re
import subprocess

import pytest

from fovpathtracing_optixcodelatest_amd import abi, lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def so():
    lib.build()
    return lib.load()


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "fovpt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fovpt_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(so):
    names = _declared_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(so, n), "include/fovpt.h declares %s but libfovpt.so does not export it" % n
    assert sorted(lib.EXPORTS) == names


def test_host_only_loader_library(so, tmp_path):
    """csrc/libfovpt_loader.so: the scene / image ingestion of the C ABI as a host-only shared object -- it exports
    every fovpt_model_* / fovpt_image_* entry point of include/fovpt.h, depends on no HIP / HSA library, and a fresh interpreter
    that only reads files through the python loaders never maps libfovpt.so or the HIP runtime."""
    path = lib.LOADER_SO_PATH
    assert os.path.exists(path), "make -C csrc builds libfovpt_loader.so beside libfovpt.so"
    L = C.CDLL(path)
    wanted = [n for n in _declared_functions() if n.startswith("fovpt_model_") or n.startswith("fovpt_image_")]
    assert len(wanted) >= 10 and sorted(wanted + ["fovpt_last_error"]) == sorted(lib.LOADER_EXPORTS)
    for n in wanted:
        assert hasattr(L, n), n
    deps = subprocess.run(["ldd", path], capture_output=True, text=True).stdout
    assert "hip" not in deps.lower() and "hsa" not in deps.lower() and "rocm" not in deps.lower(), deps
    (tmp_path / "t.ppm").write_bytes(b"P6\n2 1\n255\n" + bytes([255, 0, 0, 0, 0, 255]))
    (tmp_path / "m.mtl").write_text("newmtl m\nKd 1 1 1\nmap_Kd t.ppm\n")
    (tmp_path / "m.obj").write_text("mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nusemtl m\nf 1/1 2/1 3/1\n")
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from fovpathtracing_optixcodelatest_amd import loaders\n"
            "m = loaders.load_obj_native(%r)\n"
            "assert len(m.meshes) == 1 and len(m.textures) == 1 and m.textures[0].shape == (1, 2)\n"
            "maps = open('/proc/self/maps').read()\n"
            "assert 'libfovpt_loader.so' in maps and 'libfovpt.so' not in maps and 'libamdhip64' not in maps, maps[-2000:]\n"
            % (ROOT, str(tmp_path / "m.obj")))
    env = {k: v for k, v in os.environ.items() if k != "FOVPT_SO"}
    res = subprocess.run([os.sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert res.returncode == 0, res.stderr[-2000:]


def test_struct_layouts_match_reference_abi():
    assert C.sizeof(abi.Material) == 104          # Material.h:48-69
    assert C.sizeof(abi.Probe) == 64              # Probe.cuh:6-21
    assert C.sizeof(abi.LaunchParams) == 248      # LaunchParams.h:49-91 with nvcc alignments
    assert abi.LaunchParams.camera.offset == 104
    assert abi.LaunchParams.samples_per_launch.offset == 152
    assert abi.LaunchParams.traversable.offset == 160
    assert abi.LaunchParams.probe.offset == 168
    assert abi.LaunchParams.white.offset == 240
    assert abi.Material.transmission.offset == 80 and abi.Material.flags.offset == 100
    m = abi.Material.reference_default()          # Material.h:13-38 constructor defaults
    assert (m.transmission, m.metallic, m.eta, m.roughness) == pytest.approx((0.4, 0.5, 1.4, 1.0))
    assert m.emission.tolist() == [1.0, 1.0, 1.0] and m.color.tolist() == [1.0, 0.0, 0.0]


def test_cpp_headers_compile_and_assert_layouts():
    for src in ("shim_compile_check.cpp",):
        subprocess.check_call(["g++", "-std=c++14", "-fsyntax-only", "-I", os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "tests", "cpp", src)])
    # the plain-C view of the header must compile as C too
    subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-x", "c", "-I", os.path.join(ROOT, "include"), "-"],
                   input=b'#include "fovpt.h"\nint main(void){return sizeof(fovpt_launch_params)==248?0:1;}\n', check=True)


def test_config_mirror_matches_the_header(tmp_path):
    """abi.Config is fovpt_config member for member (the header compiled by gcc says where each member lies)."""
    names = [f[0] for f in abi.Config._fields_]
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "fovpt.h"\nint main(void){printf("%zu", sizeof(fovpt_config));' + "".join(
        'printf(" %%zu", offsetof(fovpt_config, %s));' % n for n in names) + "return 0;}\n"
    exe = str(tmp_path / "cfg_layout")
    subprocess.run(["gcc", "-std=c99", "-x", "c", "-I", os.path.join(ROOT, "include"), "-", "-o", exe], input=src.encode(), check=True)
    got = [int(x) for x in subprocess.check_output([exe]).split()]
    assert got[0] == C.sizeof(abi.Config)
    assert got[1:] == [getattr(abi.Config, n).offset for n in names]


def test_default_config_is_the_shipped_reference(so):
    c = abi.Config.reference_default()
    assert (c.uniform, c.r_inner, c.r_outer) == (0, 74, 241)                 # SimplePathtracer.cpp:20-23
    assert (c.spp_periphery, c.spp_middle, c.spp_fovea, c.spp_uniform) == (8, 16, 32, 4)   # :142,170,193,95
    assert c.max_depth == 4                                                     # deviceProgram.cu:515


def test_host_helpers_reject_bad_arguments(so):
    assert so.fovpt_probe_build_cdf(0, 4, None, None, None, None, None) != 0
    assert so.fovpt_camera_uvw(None, None, None, 45.0, 1.0, None, None, None) != 0
    assert so.fovpt_set_config(None, None) != 0
    assert so.fovpt_render(None, None) != 0


def test_fails_loudly_without_gpu(so):
    """No CPU fallback: without a HIP device the constructor raises (initOptix, SimplePathtracer.cpp:320-321)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from fovpathtracing_optixcodelatest_amd import renderer, scenes
    with pytest.raises(lib.FovptError) as e:
        renderer.SampleRenderer(scenes.cornell_box())
    assert "no HIP device" in str(e.value)


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "fovpathtracing_optixcodelatest_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle_py" not in text and "libfovpt_oracle" not in text and "orc_" not in text, f
    for f in os.listdir(os.path.join(ROOT, "include")):
        p = os.path.join(ROOT, "include", f)
        if os.path.isfile(p):
            assert "orc_" not in open(p).read()

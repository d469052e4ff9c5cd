"""CPU tests: liboatgpu.so loads and exports exactly what include/oatgpu.h declares
(no compute calls -- there is no GPU here), and the binding's struct layouts match."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(os.path.join(ROOT, "oat_amd", "lib", "liboatgpu.so")):
        subprocess.check_call(["make", "-s", "-j4", "-C", ROOT, "oat_amd/lib/liboatgpu.so"])
    from oat_amd import ffi
    return ffi.load()


def _declared():
    src = open(os.path.join(ROOT, "include", "oatgpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(oatgpu_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from oat_amd import ffi
    names = _declared()
    assert len(names) >= 24
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/oatgpu.h but not exported"
    assert sorted(ffi.SIGNATURES) == names


def test_abi_version_and_default_config(lib):
    from oat_amd import ffi
    assert lib.oatgpu_abi_version() == ffi.ABI_VERSION == 9
    cfg = ffi.Config()
    assert lib.oatgpu_default_config(C.byref(cfg)) == 0
    # cv::createBackgroundSubtractorMOG2() defaults + HSVDetector.h:77-94
    assert (cfg.history, cfg.nmixtures, cfg.detect_shadows, cfg.shadow_value) == (500, 5, 1, 127)
    assert (cfg.var_threshold, cfg.var_threshold_gen, cfg.var_init, cfg.var_min, cfg.var_max) == (16, 9, 15, 4, 75)
    assert abs(cfg.background_ratio - 0.9) < 1e-7 and abs(cfg.ct - 0.05) < 1e-8 and cfg.tau == 0.5
    assert (cfg.h_lo, cfg.h_hi, cfg.s_lo, cfg.s_hi, cfg.v_lo, cfg.v_hi) == (0, 256, 0, 256, 0, 256)
    assert (cfg.erode, cfg.dilate, cfg.min_area) == (0, 10, 0.0) and cfg.max_area > 1e308
    assert cfg.mog_restore_nmodes == 1          # MOG2Invoker: nmodes = nNewModes


def test_struct_sizes_match_the_header(lib, tmp_path):
    from oat_amd import ffi
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "oatgpu.h"\nint main(){printf("%zu %zu %zu\\n",'
                   'sizeof(oatgpu_config),sizeof(oatgpu_position),sizeof(oatgpu_profile));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert sizes == [C.sizeof(ffi.Config), C.sizeof(ffi.Position), C.sizeof(ffi.Profile)]


def test_create_without_gpu_fails_loudly_not_silently(lib):
    """No CPU fallback: with no HIP device the product refuses to run."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import oat_amd
    with pytest.raises(oat_amd.OatGpuError):
        oat_amd.HotPath(48, 64)


def test_hsv_table_integer_formula_equals_cvround():
    """The kernel builds RGB2HSV_b's tables as floor(fp32((2n+i) / (2i))): check that this equals the
    integer form and cvRound(n/i) for every entry (the fp32 quotient is exact enough for floor())."""
    import numpy as np
    for i in range(1, 256):
        for n, ref in (((255 << 12), (255 << 12) / (1.0 * i)), ((180 << 12) // 6, (180 << 12) / (6.0 * i))):
            q = int(np.floor(np.float32(2 * n + i) / np.float32(2 * i)))
            assert q == (2 * n + i) // (2 * i) == int(np.rint(ref))
            # r05: the kernels take the quotient as numerator * v_rcp_f32(denominator); the instruction is good to 1 ulp, so
            # the entry must come out right with the reciprocal one ulp off either way (and two, for margin)
            r = np.float32(1.0) / np.float32(2 * i)
            for ulps in (-2, -1, 0, 1, 2):
                rr = r
                for _ in range(abs(ulps)):
                    rr = np.nextafter(rr, np.float32(np.inf if ulps > 0 else 0), dtype=np.float32)
                assert int(np.floor(np.float32(2 * n + i) * rr)) == q, (i, n, ulps)


def test_oatgpu_lib_redirect_needs_the_measure_switch(monkeypatch):
    """OATGPU_LIB alone must not change what the Python binding loads: only together with
    OATGPU_MEASURE_PY=1, which the A/B tools under tools/ set."""
    from oat_amd import ffi
    product = os.path.join(os.path.dirname(os.path.abspath(ffi.__file__)), "lib", "liboatgpu.so")
    monkeypatch.setenv("OATGPU_LIB", "/tmp/some_other_build.so")
    monkeypatch.delenv("OATGPU_MEASURE_PY", raising=False)
    assert ffi.lib_path() == product
    monkeypatch.setenv("OATGPU_MEASURE_PY", "1")
    assert ffi.lib_path() == "/tmp/some_other_build.so"


def test_only_the_product_library_sits_beside_the_binding():
    """A/B builds live under build/variants (make variant): nothing but liboatgpu.so in oat_amd/lib."""
    libdir = os.path.join(ROOT, "oat_amd", "lib")
    assert sorted(f for f in os.listdir(libdir) if f.endswith(".so")) == ["liboatgpu.so"]

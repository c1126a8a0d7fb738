"""The C-ABI from a host that is neither Python nor PyTorch: examples/c_host_conv.c (plain C99 + the HIP runtime C API) packs a
checkpoint-layout weight with mnet_pack_weights and runs mnet_conv2d_nhwc in the fp32 and the split-half mode.
not-gpu: it compiles and links against the header and the library as C; -m gpu: it runs and agrees with its own scalar loop."""
import os
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "marconet_amd", "lib")


def _build(out):
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "c_host_conv.c"), "-L" + LIBDIR, "-lmarconet_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lm", "-o", out]
    subprocess.check_call(cmd)


def test_c_host_example_compiles_and_links_as_c99():
    from marconet_amd import _lib
    _lib.load()                                                   # the library must exist (build() made it)
    with tempfile.TemporaryDirectory(prefix="mnet_c_host_") as d:
        exe = os.path.join(d, "c_host_conv")
        _build(exe)
        assert os.path.getsize(exe) > 0


@pytest.mark.gpu
def test_c_host_example_runs():
    with tempfile.TemporaryDirectory(prefix="mnet_c_host_") as d:
        exe = os.path.join(d, "c_host_conv")
        _build(exe)
        env = dict(os.environ, LD_LIBRARY_PATH=LIBDIR + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
        r = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=300)
        print(r.stdout, r.stderr)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "c_host_conv: max-abs" in r.stdout

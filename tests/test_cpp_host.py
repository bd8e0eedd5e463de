"""The C++ host mirror (include/svo_render.hpp: Gpu / Render / Compute / Octree / CpuOctree with the
reference's method names) compiles with plain g++ against the C ABI -- no HIP or torch types cross it --
and the example host program runs."""
import os
import subprocess

import pytest

from conftest import ROOT


def _build(tmp_path, pkg, name="render_frame"):
    exe = str(tmp_path / name)
    libdir = os.path.dirname(pkg._lib.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", name + ".cpp"), "-L", libdir, "-lsvo_hip",
                           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def test_cpp_host_compiles_and_builds_tree(tmp_path, pkg):
    exe = _build(tmp_path, pkg)
    out = subprocess.run([exe, "--host-only"], capture_output=True, text=True, check=True).stdout
    assert "tree: 456 words" in out  # 8 * (1 + 56 interior nodes)
    assert "world: 920 words fully expanded" in out  # root group + two instances of the 456-word block


@pytest.mark.gpu
def test_cpp_host_renders(tmp_path, pkg, gpu):
    exe = _build(tmp_path, pkg)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "hits" in r.stdout and "scan:" in r.stdout and "streaming:" in r.stdout


def test_cpp_multi_gpu_host_compiles(tmp_path, pkg):
    """the single-process multi-GPU host (svo::MultiGpuFrame: comm_init_all + one gather per frame) builds with plain g++"""
    _build(tmp_path, pkg, "multi_gpu_frame")


@pytest.mark.gpu
def test_cpp_multi_gpu_host_runs_with_the_visible_gpus(tmp_path, pkg, gpu):
    """... and its sharded frame (tiles -> 12-byte wire records -> RCCL gather behind the C ABI -> assemble) equals the
    unsharded frame, with as many ranks as the box has GPUs (one on the test box: the gather is then a device copy)."""
    exe = _build(tmp_path, pkg, "multi_gpu_frame")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "sharded frame equals the unsharded one" in r.stdout

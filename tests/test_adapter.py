"""The C++ drop-in adapter (include/calibba_adapter.hpp) — SURVEY.md §8(b).

CPU tier: the header must compile as C++20 with every templated entry point instantiated for the two camera types the
reference instantiates (intrinsics.cpp:122-132, extrinsics.cpp:198-207, bundle.cpp:172-179), against the TEST-ONLY stand-in
declarations of the Eigen / calib:: types it touches (tests/adapter_check/stand_ins/; they check the adapter's syntax and
types and pin nothing), and the test program must link against libcalibba.so.
GPU tier: the program drives every adapter function on noise-free scenes (ground-truth recovery, result shapes, and the
exception types of the reference's argument checks).
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIR = os.path.join(ROOT, "tests", "adapter_check")
EXE = os.path.join(DIR, "_build", "adapter_drive")


def test_adapter_header_is_valid_cpp20_for_both_camera_types():
    cmd = ["g++", "-std=c++20", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(DIR, "stand_ins"),
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "oracle"), os.path.join(DIR, "adapter_drive.cpp")]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr


def test_adapter_program_links_against_libcalibba():
    p = subprocess.run(["make", "-C", DIR], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_adapter_drives_every_entry_point_on_the_gpu():
    if not os.path.exists(EXE):
        subprocess.run(["make", "-C", DIR], check=True)
    p = subprocess.run([EXE], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "adapter_drive: all ok" in p.stdout, p.stdout

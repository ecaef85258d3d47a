"""One error convention for all five shim files ("void/int returns, no exceptions"; shim/shim_error.h): with NO usable device every
replaced member function of the drop-in library returns the reference's "nothing found" value, counts the failure in the process-wide channel
(orbx_shim_error_count / orbx_shim_last_error) and does not throw - an exception escaping into the reference's callers would end the process
(std::terminate on the LocalMapping thread, src/LocalMapping.cc:123; here: through the C wrapper).  The calls run in a child process whose HIP
runtime sees no device (works on the GPU box too); ORBX_SHIM_FATAL=1 turns the same failure into a std::runtime_error = an aborted child."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

import oracle_lib

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.skipif(oracle_lib.slam_lib() is None or not oracle_lib.SLAM_HIP_SO.exists(),
                                reason="oracle/_ref/liborbslam{,_hip}.so not built (needs /root/reference)")

CHILD = r'''
import ctypes, importlib, sys
import numpy as np
sys.path.insert(0, %(tests)r); sys.path.insert(0, %(root)r)
import oracle_lib
orbx = importlib.import_module("self_commit_orb-slam2_amd")
hip = oracle_lib.slam_hip_lib()
hip.orbx_shim_error_count.restype = ctypes.c_long
hip.orbx_shim_last_error.argtypes = [ctypes.c_char_p, ctypes.c_int]
def errs(): return hip.orbx_shim_error_count()
def last():
    b = ctypes.create_string_buffer(512); hip.orbx_shim_last_error(b, 512); return b.value.decode()
seen = {}
def step(name, fn, check):
    before = errs()
    r = fn()
    assert errs() > before, name + ": the failure was not counted"
    assert check(r), (name, r)
    seen[name] = last()
    print("ok", name, "|", seen[name][:90], flush=True)
from test_matcher import _noisy_pair
rng = np.random.default_rng(1)
kA, dA, kB, dB = _noisy_pair(rng, 300, orbx)
# shim/ORBmatcher_hip.cc
step("SearchByBoW(KF,F)", lambda: oracle_lib.ref_search_by_bow(0, kA, dA, kB, dB, 0.7, True, lib=hip), lambda r: r[0] == 0 and (r[1] < 0).all())
step("SearchByBoW(KF,KF)", lambda: oracle_lib.ref_search_by_bow(1, kA, dA, kB, dB, 0.7, True, lib=hip), lambda r: r[0] == 0 and (r[1] < 0).all())
from test_search_init import _frames
f1, f2, prev = _frames(orbx, 3, 400)
step("SearchForInitialization", lambda: oracle_lib.ref_search_for_initialization(f1, f2, prev, 30, 0.9, True, lib=hip), lambda r: r[0] == 0)
# shim/ORBextractor.cc + shim/Frame_hip.cc: the monocular constructor on a distorted camera (extractor, UndistortKeyPoints, ComputeImageBounds, AssignFeaturesToGrid)
im = orbx.synth_frame(5, 640, 480)
hip.orbslam_keep_frame_statics(0)
step("Frame(mono)", lambda: oracle_lib.ref_mono_frame(im, 1000, 517.3, 516.5, 318.6, 255.3, (0.262383, -0.953104, -0.005358, 0.002628, 1.163314), lib=hip),
     lambda r: len(r["kps"]) == 0 and len(r["kpsUn"]) == 0 and r["bounds"][0] == 0.0 and r["bounds"][1] == 640.0 and r["gridOff"][-1] == 0)
imR = orbx.synth_frame(5, 640, 480, orbx.SYNTH_STEREO_RIGHT)
step("Frame(stereo)", lambda: oracle_lib.ref_stereo_frame(im, imR, 1000, 500.0, 500.0, 320.0, 240.0, 40.0, lib=hip), lambda r: len(r["kpsL"]) == 0 and len(r["uRight"]) == 0)
# shim/Optimizer_hip.cc
w = orbx.lba_synth.make_window(K=8, P=200, seed=2, n_fixed=0)
step("LocalBundleAdjustment", lambda: oracle_lib.ref_local_ba_on_map(w, 7, lib=hip), lambda r: (r["poses"] == w["poses"]).all() and r["erased"].sum() == 0)
from test_pose_optimization import make_frame
fr = make_frame(4)
step("PoseOptimization", lambda: oracle_lib.ref_pose_optimization_on_frame(fr, lib=hip), lambda r: r["inliers"] == 0 and (np.asarray(r["pose"]) == np.asarray(fr["pose"], np.float32)).all())
print("DONE", errs(), flush=True)
'''


def _run(extra_env):
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1", **extra_env)
    code = CHILD % {"tests": str(ROOT / "tests"), "root": str(ROOT)}
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))


def test_no_shim_entry_throws_without_a_device():
    r = _run({})
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "DONE" in r.stdout
    for name in ("SearchByBoW(KF,F)", "SearchByBoW(KF,KF)", "SearchForInitialization", "Frame(mono)", "Frame(stereo)", "LocalBundleAdjustment", "PoseOptimization"):
        assert "ok " + name in r.stdout, (name, r.stdout[-1500:])
    assert "(orbx)" in r.stderr      # the failures were also written to std::cerr


def test_fatal_mode_turns_the_first_failure_into_an_exception():
    r = _run({"ORBX_SHIM_FATAL": "1"})
    assert r.returncode != 0 and "DONE" not in r.stdout, (r.returncode, r.stdout[-500:])

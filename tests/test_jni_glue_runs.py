"""csrc/jni_glue.cpp EXECUTED: tests/jni_stub/fake_jvm.cpp stands in for the JVM (the JNIEnv members of the test-only jni.h over
fake Java arrays, a RunListener, pending exceptions — copying Get/Release semantics, so a wrong release mode loses the data) and
drives the glue's natives the way java/.../HipSceneLoader, HipPathTracingRenderer and HipPreviewRenderer do.  No JDK exists in
this image, so this is as close to running the Java side as the repository gets; what a real JVM adds (the JNI function-table
ABI, threads, GC) stays unverified."""
import json
import os
import subprocess

import numpy as np
import pytest

import golden_scenes as gs
from chunkyclplugin_amd import native, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fake_jvm(tmp_path_factory):
    native.lib()   # (builds the library only where it is missing)
    exe = str(tmp_path_factory.mktemp("jvm") / "fake_jvm")
    cmd = ["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "tests", "jni_stub"),
           os.path.join(ROOT, "tests", "jni_stub", "fake_jvm.cpp"), os.path.join(native.CSRC, "jni_glue.cpp"),
           "-o", exe, "-L" + native.PKG_DIR, "-lchunky_hip", "-Wl,-rpath," + native.PKG_DIR, "-Wl,--allow-shlib-undefined"]
    proc = subprocess.run(cmd, capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr
    return exe


def run(exe, sc, tmp_path, target, interval, *extra):
    raw, out = str(tmp_path / "scene.raw"), str(tmp_path / "out.f64")
    scenes.save_raw(sc, raw)
    proc = subprocess.run([exe, raw, out, str(target), str(interval), *[str(x) for x in extra]], capture_output=True, text=True, timeout=300)
    lines = [json.loads(x) for x in proc.stdout.strip().splitlines() if x.startswith("{")]
    return proc, lines, out


def test_glue_builds_and_reports_a_missing_gpu_as_a_java_exception(fake_jvm, tmp_path):
    sc = scenes.tiny_scene(width=48, height=32)
    proc, lines, out = run(fake_jvm, sc, tmp_path, 4, 2)
    if native.lib().chunky_device_count() > 0:
        assert proc.returncode == 0, (proc.stdout, proc.stderr)
        return
    # HipNative.init -> RuntimeException carrying chunky_last_error(); nothing after it runs
    assert proc.returncode == 3 and lines[0] == {"devices": 0}
    assert lines[1]["exception_at"] == "init" and lines[1]["class"] == "java/lang/RuntimeException" and "no HIP device" in lines[1]["message"]
    assert not os.path.exists(out)


def oracle_loop(port, sc, merges):
    """the reference's loop through the given merge points (spp after each merge)"""
    seeds = scenes.java_random_ints(merges[-1])
    want = np.zeros(sc.width * sc.height * 3, np.float64)
    done = 0
    for upto in merges:
        m = upto - done
        pass_buf = port.render_passes(sc, seeds[done:upto]).astype(np.float64)
        want = (want * done + pass_buf * m) * (1.0 / upto)
        done = upto
    return want


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["indoor", "entities"])
def test_glue_runs_the_renderer_classes_scenario(fake_jvm, tmp_path, port, name):
    sc = gs.make(name).with_view(64, 48)
    proc, lines, out = run(fake_jvm, sc, tmp_path, 10, 4)
    assert proc.returncode == 0, (proc.stdout, proc.stderr)
    s = lines[-1]
    assert s["violations"] == 0 and s["unreleased_arrays"] == 0 and s["thrown"] == 0
    assert s["spp"] == 10 and s["merges"] == [4, 8, 10] and s["progress_monotonic"] and s["progress_calls"] >= 3 and s["regenerate_calls"] == s["progress_calls"]
    assert s["array_nonzero_at_first_merge"]  # merged() saw the samples in the Java array, not only in the native buffer
    assert any("string" in x and "gfx950" in x["string"] for x in lines)  # deviceName -> NewStringUTF
    got = np.fromfile(out, np.float64)
    np.testing.assert_array_equal(got.view(np.uint64), oracle_loop(port, sc, [4, 8, 10]).view(np.uint64))
    # the int[] of renderPreview and the float[] of renderRead arrived in the "Java" arrays (release mode 0)
    assert s["preview_sum"] == int(port.preview(sc).astype(np.uint32).astype(np.uint64).sum())
    one = port.render_passes(sc, scenes.java_random_ints(1)).astype(np.float64)
    assert abs(s["read_sum"] - float(one.sum())) <= 1e-9 * abs(float(one.sum()))


@pytest.mark.gpu
def test_glue_save_event_and_stop(fake_jvm, tmp_path, port):
    sc = gs.make("indoor").with_view(48, 32)
    # a snapshot due at 7 spp cuts the launch there and merges at once (OpenClPathTracingRenderer.java:150-151)
    proc, lines, out = run(fake_jvm, sc, tmp_path, 10, 4, "save-at", 7)
    assert proc.returncode == 0, (proc.stdout, proc.stderr)
    assert lines[-1]["merges"] == [4, 7, 10] and lines[-1]["spp"] == 10 and lines[-1]["violations"] == 0
    np.testing.assert_array_equal(np.fromfile(out, np.float64).view(np.uint64), oracle_loop(port, sc, [4, 7, 10]).view(np.uint64))
    # postRender returning true at its first poll stops the loop: no exception, fewer samples than asked for
    proc, lines, out = run(fake_jvm, sc, tmp_path, 100000, 1024, "stop-after-polls", 0)
    assert proc.returncode == 0, (proc.stdout, proc.stderr)
    s = lines[-1]
    assert s["polls"] >= 1 and s["spp"] < 100000 and s["thrown"] == 0 and s["violations"] == 0 and s["unreleased_arrays"] == 0


@pytest.mark.gpu
def test_glue_guards_the_java_heap(fake_jvm, tmp_path, port):
    """arrays too short for what the C side would read or write raise IllegalArgumentException BEFORE the C call; C-side refusals
    (wrong length for the target, NULL handle) arrive as RuntimeException; the session goes on afterwards"""
    sc = gs.make("indoor").with_view(48, 32)
    proc, lines, out = run(fake_jvm, sc, tmp_path, 6, 4, "short-arrays")
    assert proc.returncode == 0, (proc.stdout, proc.stderr)
    exc = {x["exception_at"]: x for x in lines if "exception_at" in x}
    for where in ("short sun", "short sky", "short preview", "short sample buffer"):
        assert exc[where]["class"] == "java/lang/IllegalArgumentException", exc
    for where in ("wrong-length read", "null scene"):
        assert exc[where]["class"] == "java/lang/RuntimeException" and exc[where]["message"], exc
    s = lines[-1]
    assert s["thrown"] == 6 and s["violations"] == 0 and s["unreleased_arrays"] == 0 and s["merges"] == [4, 6]
    np.testing.assert_array_equal(np.fromfile(out, np.float64).view(np.uint64), oracle_loop(port, sc, [4, 6]).view(np.uint64))

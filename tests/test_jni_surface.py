"""SURVEY.md section 8 row f3 without a JDK: the JNI layer a maintainer builds (java/.../HipNative.java +
chunkyclplugin_amd/csrc/jni_glue.cpp) is checked as far as this image allows —

* every `native` method declared in HipNative.java has an export
  `Java_dev_thatredox_chunkynative_hip_HipNative_<name>` in the glue with the same parameter count and JNI types, and
  the glue exports nothing HipNative does not declare (a missing binding is an UnsatisfiedLinkError at run time);
* the glue passes `g++ -fsyntax-only -Wall -Wextra` against tests/jni_stub/jni.h, a test-only header that only declares
  the JNI types and JNIEnv members of the JNI specification (nothing is linked or run);
* every C function the glue calls is declared in include/chunky_hip.h, and the callbacks it installs are the members of
  chunky_run_callbacks;
* the Java sources are at least brace-balanced and every HipNative.<method>( call elsewhere under java/ names a
  declared method with the declared number of arguments.
"""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JAVA = os.path.join(ROOT, "java", "dev", "thatredox", "chunkynative", "hip")
GLUE = os.path.join(ROOT, "chunkyclplugin_amd", "csrc", "jni_glue.cpp")
HEADER = os.path.join(ROOT, "include", "chunky_hip.h")

JNI_TYPE = {"int": "jint", "long": "jlong", "float": "jfloat", "double": "jdouble", "boolean": "jboolean", "void": "void",
            "int[]": "jintArray", "byte[]": "jbyteArray", "float[]": "jfloatArray", "double[]": "jdoubleArray",
            "String": "jstring"}


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def java_natives():
    src = strip_comments(open(os.path.join(JAVA, "HipNative.java")).read())
    out = {}
    for ret, name, args in re.findall(r"public\s+static\s+native\s+([\w\[\].]+)\s+(\w+)\s*\(([^)]*)\)\s*;", src):
        params = [a.strip().rsplit(" ", 1)[0].strip() for a in args.split(",") if a.strip()]
        out[name] = (ret, params)
    return out


def glue_exports():
    src = strip_comments(open(GLUE).read())
    out = {}
    for ret, name, args in re.findall(r"JNIEXPORT\s+(\w+)\s+JNICALL\s+J\((\w+)\)\s*\(([^)]*)\)", src):
        params = [a.strip() for a in args.split(",")]
        assert params[0].startswith("JNIEnv*") and params[1].startswith("jclass"), name
        out[name] = (ret, [p.split()[0] for p in params[2:]])
    return out


def test_every_native_method_has_a_glue_export_with_the_same_signature():
    natives, exports = java_natives(), glue_exports()
    assert len(natives) >= 20
    assert sorted(natives) == sorted(exports), (set(natives) ^ set(exports))
    for name, (ret, params) in natives.items():
        eret, eparams = exports[name]
        want = [JNI_TYPE.get(p, "jobject") for p in params]   # interfaces / objects travel as jobject
        assert eret == JNI_TYPE[ret], (name, ret, eret)
        assert eparams == want, (name, params, eparams)


def test_glue_compiles_against_the_jni_specification_surface():
    p = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "tests", "jni_stub"), GLUE],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    # and without the stub the product build sees an empty translation unit (no jni.h in this image)
    q = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", GLUE], capture_output=True, text=True)
    assert q.returncode == 0, q.stderr[-2000:]


def test_glue_calls_only_declared_c_functions():
    header = strip_comments(open(HEADER).read())
    declared = set(re.findall(r"\b(chunky_[a-z_0-9]+)\s*\(", header))
    used = set(re.findall(r"\b(chunky_[a-z_0-9]+)\s*\(", strip_comments(open(GLUE).read())))
    assert used <= declared, used - declared
    fields = re.search(r"typedef struct chunky_run_callbacks \{(.*?)\}", header, flags=re.S).group(1)
    members = re.findall(r"\(\*(\w+)\)", fields)
    assert members == ["post_render", "progress", "merged", "save_event", "regenerate_camera", "poll_gate"]
    glue = open(GLUE).read()
    for m in members:
        assert f"cb_{m}" in glue, m
    # the listener method names / descriptors the glue looks up are the ones HipNative.RunListener declares
    native_src = strip_comments(open(os.path.join(JAVA, "HipNative.java")).read())
    for name, sig in re.findall(r'GetMethodID\(cls, "(\w+)", "([^"]+)"\)', glue):
        ret = {"Z": "boolean", "V": "void", "I": "int"}[sig[-1]]
        arg = "int \\w+" if "(I)" in sig else ""
        assert re.search(rf"{ret}\s+{name}\s*\(\s*{arg}\s*\)\s*;", native_src), (name, sig)


def test_java_sources_are_consistent_with_hipnative():
    natives = java_natives()
    for fn in sorted(os.listdir(JAVA)):
        src = strip_comments(open(os.path.join(JAVA, fn)).read())
        src_nostr = re.sub(r'"(\\.|[^"\\])*"', '""', src)
        assert src_nostr.count("{") == src_nostr.count("}"), fn
        assert src_nostr.count("(") == src_nostr.count(")"), fn
        for m in re.finditer(r"HipNative\.(\w+)\s*\(", src_nostr):
            name = m.group(1)
            if name not in natives:
                continue                     # constants, nested types
            depth, i, n_args, seen = 1, m.end(), 0, False
            while depth:
                c = src_nostr[i]
                if c in "([{":
                    depth += 1
                elif c in ")]}":
                    depth -= 1
                elif c == "," and depth == 1:
                    n_args += 1
                if depth and not c.isspace():
                    seen = True
                i += 1
            n_args = n_args + 1 if seen else 0
            assert n_args == len(natives[name][1]), (fn, name, n_args, natives[name][1])

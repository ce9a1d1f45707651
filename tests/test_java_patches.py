"""The three reference classes that keep their logic (ClTextureLoader, ClSky, ClCamera) are PATCHED, not re-written: the edit
list java/patches/edits.json (line numbers + the new lines, no reference text) and java/patches/apply_edits.py.  Here the
edits are applied to a temporary copy of the reference's own files (only where /root/reference exists) and the result is
checked as far as this image allows (no JDK): no JOCL left, braces balanced, every HipNative call declared with that many
arguments, and the constructors / methods the Hip* classes under java/ call on them exist with those arities — so every
caller of HipNative.sceneSetAtlas / sceneWriteAtlasTile / sceneSetSky / renderSetCamera is under test again."""
import importlib.util
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATCHES = os.path.join(ROOT, "java", "patches")
JAVA = os.path.join(ROOT, "java", "dev", "thatredox", "chunkynative", "hip")
REFERENCE = "/root/reference"

spec = importlib.util.spec_from_file_location("apply_edits", os.path.join(PATCHES, "apply_edits.py"))
apply_edits = importlib.util.module_from_spec(spec)
spec.loader.exec_module(apply_edits)
EDITS = json.load(open(os.path.join(PATCHES, "edits.json")))


def strip(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return re.sub(r'"(?:\\.|[^"\\])*"', '""', text)


def natives():
    src = strip(open(os.path.join(JAVA, "HipNative.java")).read())
    return {name: len([a for a in args.split(",") if a.strip()])
            for name, args in re.findall(r"public\s+static\s+native\s+[\w\[\].]+\s+(\w+)\s*\(([^)]*)\)\s*;", src)}


def call_arities(src, pattern):
    """argument counts of every call matching `pattern(` in src (nested parentheses counted)."""
    out = []
    for m in re.finditer(pattern + r"\s*\(", src):
        depth, args, i, seen = 1, 1, m.end(), False
        while depth and i < len(src):
            c = src[i]
            if c in "([{":
                depth += 1
            elif c in ")]}":
                depth -= 1
            elif c == "," and depth == 1:
                args += 1
            if depth and not c.isspace():
                seen = True
            i += 1
        out.append(args if seen else 0)
    return out


def test_edit_list_is_well_formed():
    assert [f["path"].rsplit("/", 1)[1] for f in EDITS["files"]] == ["ClTextureLoader.java", "ClSky.java", "ClCamera.java"]
    for f in EDITS["files"]:
        assert re.fullmatch(r"[0-9a-f]{64}", f["sha256"])
        last = 0
        for e in f["edits"]:
            assert e["first"] > last and e["last"] >= e["first"] - 1 and isinstance(e["with"], list) and e["why"]
            last = max(e["last"], e["first"] - 1)
        # the edit list stores only what is NEW: nothing of JOCL may be among the lines it writes
        new = "\n".join(line for e in f["edits"] for line in e["with"])
        assert not re.search(r"org\.jocl|clCreate|clEnqueue|ClMemory|cl_mem|RendererInstance", new)


def test_apply_file_semantics():
    lines = [f"l{i}" for i in range(1, 9)]
    out = apply_edits.apply_file(lines, [{"first": 2, "last": 3, "with": ["A"]}, {"first": 5, "last": 4, "with": ["B", "C"]}, {"first": 8, "last": 8, "with": []}])
    assert out == ["l1", "A", "l4", "B", "C", "l5", "l6", "l7"]
    with pytest.raises(ValueError):
        apply_edits.apply_file(lines, [{"first": 2, "last": 4, "with": []}, {"first": 4, "last": 5, "with": []}])
    with pytest.raises(ValueError):
        apply_edits.apply_file(lines, [{"first": 7, "last": 9, "with": []}])


@pytest.fixture(scope="module")
def patched(tmp_path_factory):
    if not os.path.isdir(REFERENCE):
        pytest.skip("needs the reference checkout (/root/reference)")
    out = str(tmp_path_factory.mktemp("patched"))
    assert apply_edits.main([REFERENCE, "--out", out]) == 0
    files = {}
    for f in EDITS["files"]:
        files[f["path"].rsplit("/", 1)[1][:-5]] = open(os.path.join(out, EDITS["root"], f["path"])).read()
    return files


def test_reference_files_are_the_pinned_version():
    if not os.path.isdir(REFERENCE):
        pytest.skip("needs the reference checkout (/root/reference)")
    assert apply_edits.main([REFERENCE, "--check"]) == 0


def test_patched_classes_are_free_of_jocl_and_balanced(patched):
    for name, src in patched.items():
        code = strip(src)
        assert not re.search(r"org\.jocl|\bcl[A-Z]\w*\(|\bCL_[A-Z_]+\b|ClMemory|cl_mem|cl_image|RendererInstance|Sizeof|Pointer\.to|AutoCloseable|@Override\s+public void close", code), name
        for a, b in ("{}", "()", "[]"):
            assert code.count(a) == code.count(b), (name, a)
        assert "import dev.thatredox.chunkynative.hip.HipNative;" in src
        assert re.search(r"public class %s\b" % name, code)


def test_patched_classes_call_declared_natives(patched):
    nat = natives()
    used = set()
    for name, src in patched.items():
        code = strip(src)
        for m in set(re.findall(r"HipNative\.(\w+)\s*\(", code)):
            assert m in nat, (name, m)
            for n in call_arities(code, r"HipNative\." + m):
                assert n == nat[m], (name, m, n, nat[m])
            used.add(m)
    assert used == {"sceneSetAtlas", "sceneWriteAtlasTile", "sceneSetSky", "renderSetCamera"}


def test_hip_classes_call_the_patched_surface(patched):
    """What java/.../hip/*.java expects of the patched classes is what the patches produce."""
    sky, tex, cam = strip(patched["ClSky"]), strip(patched["ClTextureLoader"]), strip(patched["ClCamera"])
    assert re.search(r"public ClSky\(long \w+, Scene \w+\)", sky)
    assert re.search(r"public ClTextureLoader\(long \w+\)", tex)
    assert re.search(r"public ClCamera\(Scene \w+\)", cam) and re.search(r"public void apply\(long \w+\)", cam)
    assert re.search(r"public void generate\(long \w+, boolean \w+\)", cam)
    callers = "\n".join(strip(open(os.path.join(JAVA, f)).read()) for f in sorted(os.listdir(JAVA)) if f.endswith(".java"))
    assert call_arities(callers, r"new ClSky") == [2]
    assert call_arities(callers, r"new ClTextureLoader") == [1]
    assert set(call_arities(callers, r"new ClCamera")) == {1}
    assert set(call_arities(callers, r"camera\.apply")) == {1}
    assert set(call_arities(callers, r"camera\.generate")) == {2}


def test_java_sources_lex_cleanly(patched):
    """No JDK here: the closest thing to a compiler front end in the image is Pygments' Java lexer.  Every file under java/ and the
    three patched classes must lex without a single error token, every string / char literal must close, and every
    `import dev.thatredox.chunkynative.hip.X` must name a class that exists under java/."""
    pygments = pytest.importorskip("pygments")
    from pygments.lexers import JavaLexer
    from pygments.token import Error
    sources = {f: open(os.path.join(JAVA, f)).read() for f in sorted(os.listdir(JAVA)) if f.endswith(".java")}
    sources.update({k + ".java (patched)": v for k, v in patched.items()})
    have = {f[:-5] for f in os.listdir(JAVA) if f.endswith(".java")}
    for name, src in sources.items():
        bad = [v for t, v in JavaLexer().get_tokens(src) if t is Error]
        assert not bad, (name, bad[:5])
        for cls in re.findall(r"import dev\.thatredox\.chunkynative\.hip\.(\w+);", src):
            assert cls in have, (name, cls)
        assert re.search(r"^package dev\.thatredox\.chunkynative\.", src, flags=re.M), name

#!/usr/bin/env python3
"""tools/isa_scratch.py [-D...] [--kernel MANGLED-SUBSTRING] — where the scratch (spill) instructions of a render_pool instantiation sit.

Compiles csrc/render_pool.hip for gfx950 with the library's flags (+ any -D given), cuts the instantiation's ISA out of hipcc's
assembly and attributes every `scratch_` instruction to the phase of the state machine its basic block belongs to.  Phases are
found by landmarks no other phase contains: the march loops hold the `v_cvt_flr_i32_f32` of march_step (inline asm), SHADE the
non-temporal staging store and the sample-claim atomic, the swap the fourteen `ds_wrxchg_rtn_b64`; BLOCK is what lies between
SHADE and the march loops, the model blocks' phase what lies between two comments render_pool.hip leaves in the compiled kernel.  Output: one line per scratch instruction (line, basic block, loop depth
as the assembler comments state it, phase) and a count per phase.  CPU only (hipcc cross-compiles)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chunkyclplugin_amd import native  # noqa: E402

args = sys.argv[1:]
kernel = "render_poolILi17ELi64ELb0ELb0ELb0ELb0EE"  # (render_pool<17, 64, false, false, false, false>; the last flag: sorted block tests)
if "--kernel" in args:
    i = args.index("--kernel")
    kernel = args[i + 1]
    del args[i:i + 2]
flags = [f for f in native.HIPCC_FLAGS if f != "-shared"] + args
with tempfile.TemporaryDirectory() as td:
    subprocess.run(["hipcc", *flags, "-x", "hip", "-c", os.path.join(native.CSRC, "render_pool.hip"), "-o", os.path.join(td, "rp.o"), "--save-temps"],
                   cwd=td, check=True, capture_output=True)
    text = open(os.path.join(td, "render_pool-hip-amdgcn-amd-amdhsa-gfx950.s")).read().splitlines()
start = next(i for i, ln in enumerate(text) if ln.startswith("_ZN6chunky") and kernel in ln and ln.rstrip().split(":")[0].endswith("E") and ":" in ln)
end = next(i for i in range(start, len(text)) if ".end_amdhsa_kernel" in text[i])
body = text[start:end]
tail = text[end:end + 60]  # the resource comments follow the kernel descriptor
meta = {k: next((re.search(r"(\d+)", ln.split(k)[1]).group(1) for ln in tail if k in ln), None)
        for k in ("; NumVgprs:", "; ScratchSize:", "; Occupancy:", "; TotalNumSgprs:", "; codeLenInByte =")}

# phase landmarks -> line ranges.  The kernel's main loop is one big loop; its phases are contiguous regions in program order.
marks = []
for i, ln in enumerate(body):
    if "v_cvt_flr_i32_f32" in ln:
        marks.append((i, "MARCH"))
    elif "ds_wrxchg_rtn_b64" in ln:
        marks.append((i, "SWAP"))
    elif re.search(r"global_store_dword.* nt", ln) or "global_atomic_add" in ln:
        marks.append((i, "SHADE"))
    elif "s_getreg_b32" in ln:
        marks.append((i, "PROLOGUE"))
    elif "chunky-mark models-end" in ln:
        marks.append((i, "MODELS-END"))
    elif "chunky-mark models" in ln:
        marks.append((i, "MODELS"))
# BLOCK: everything between the last SHADE landmark and the first MARCH landmark (program order of the compiled kernel)
last_shade = max((i for i, p in marks if p == "SHADE"), default=0)
first_march = min((i for i, p in marks if p == "MARCH"), default=len(body))
first_swap = min((i for i, p in marks if p == "SWAP"), default=0)
last_swap = max((i for i, p in marks if p == "SWAP"), default=0)
# (render_pool.hip leaves a comment where the cube test ends and the model-block tests begin)
first_models = min((i for i, p in marks if p == "MODELS"), default=None)
last_models = max((i for i, p in marks if p == "MODELS-END"), default=None)


def phase_of(i):
    if i < first_swap - 250:
        return "PROLOGUE (before the main loop)"
    if i <= last_swap + 60:
        return "VOTE+SWAP"
    if first_models is not None and first_models - 40 <= i <= last_models + 40:  # (the phase's code may sit anywhere in program order)
        return "MODEL (model blocks)"
    if i <= last_shade + 40:
        return "SHADE (incl. new samples, trace_setup)"
    if i < first_march - 120:
        return "BLOCK"
    return "MARCH"


label, depth = "(entry)", "0"
counts = {}
print(f"# {text[start].split(':')[0]}")
print(f"# VGPRs {meta['; NumVgprs:']}  SGPRs {meta['; TotalNumSgprs:']}  scratch bytes/lane {meta['; ScratchSize:']}  waves/SIMD {meta['; Occupancy:']}  code bytes {meta['; codeLenInByte =']}  flags: {' '.join(args) or '(library defaults)'}")
print(f"# landmarks (ISA line of the kernel): swap {first_swap}-{last_swap}, SHADE's staging store / claim up to {last_shade}, march loops from {first_march}; {len(body)} lines")
for i, ln in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):\s*;?(.*)", ln)
    if m:
        label = m.group(1)
        d = re.search(r"Depth=(\d+)", m.group(2))
        depth = d.group(1) if d else "?"
    if "scratch_" in ln:
        ph = "PROLOGUE (before the main loop)" if depth == "0" else phase_of(i)
        counts[ph] = counts.get(ph, 0) + 1
        print(f"{i:5d}  {label:12s} depth {depth}  {ph:40s} {ln.strip()}")
print("# per phase:", ", ".join(f"{k}: {v}" for k, v in sorted(counts.items())) or "no scratch instructions")
print("# in the march loops:", counts.get("MARCH", 0))

#!/usr/bin/env python3
"""What hipcc made of the fast kernels: resource usage per kernel, and proof that every instance of the generated symbol loop sits in
the code object exactly as generated -- instruction for instruction, nothing of the compiler's (in particular no scratch_ spill
traffic) between the loop's first and last instruction.

    python3 tools/check_code_object.py [lzma_rs_amd/libmilzma.so]

The loop is one inline-asm statement per variant, so the compiler cannot touch its inside; this checks that belief against the
built binary (the time-sliced instantiation spills hundreds of VGPRs around the loop).  Used by
tests/test_host_abi.py::test_asm_loops_sit_in_the_code_object_untouched."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
sys.path.insert(0, os.path.join(ROOT, "tools"))


def extract(so, workdir):
    fat, co = os.path.join(workdir, "fat.bin"), os.path.join(workdir, "k.co")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, so, os.path.join(workdir, "ignored.so")])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    return co


def kernel_metadata(co):
    notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
    out, cur = {}, {}
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s+(.*)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count" and cur.get("name"):   # (first key of a kernel's map in the note: the previous one is complete)
            out[cur["name"]] = cur
            cur = {}
        if k in ("name", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
                 "group_segment_fixed_size", "agpr_count"):
            cur[k] = v if k == "name" else int(v)
    if cur.get("name"):
        out[cur["name"]] = cur
    return out


def norm(mnemonic):
    return re.sub(r"_(e32|e64|sdwa|dpp)$", "", mnemonic)


def disassemble(co):
    """-> {function name: [mnemonic, ...]} (labels inside a function dropped)"""
    text = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], text=True, stderr=subprocess.DEVNULL)
    funcs, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\w+)>:", line)
        if m:
            if m.group(1).startswith("_Z"):
                cur = funcs.setdefault(m.group(1), [])
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\b", line)
        if m and cur is not None:
            cur.append(norm(m.group(1)))
    return funcs


def generated_variants():
    import gen_fast_loop as G
    out = {}
    for name, lp0, pb4 in (("LP0", True, False), ("GEN", False, False), ("PB4", False, True), ("HBM", False, True), ("HB0", True, False)):
        g = G.Gen(lp0, pb4, hbm=(name in ("HBM", "HB0")))
        g.build()
        lines = g.main + g.cold + g.cold2 + g.stubs
        g.cur = lines
        g.finish()
        out[name] = [norm(l.split()[0]) for l in lines if not l.strip().endswith(":")]
    return out


def check(so):
    with tempfile.TemporaryDirectory() as wd:
        co = extract(so, wd)
        meta = kernel_metadata(co)
        funcs = disassemble(co)
    variants = generated_variants()
    report = {"kernels": {}, "loops": []}
    for name, m in meta.items():
        if "decode_fast_asm" in name:
            report["kernels"][name] = m
    problems = []
    for fname, mn in funcs.items():
        if "decode_fast_asm" not in fname:
            continue
        want = ["LP0", "GEN", "PB4", "HBM", "HB0"]
        anchors = [i for i, x in enumerate(mn) if x == "s_getpc_b64"]
        found = []
        for a in anchors:
            hit = None
            for v in want:
                gen = variants[v]
                k = gen.index("s_getpc_b64")
                if a - k >= 0 and mn[a - k:a - k + len(gen)] == gen:
                    hit = v
                    break
            if hit:
                found.append(hit)
                lo, hi = a - variants[hit].index("s_getpc_b64"), a - variants[hit].index("s_getpc_b64") + len(variants[hit])
                assert not any(x.startswith("scratch_") for x in mn[lo:hi])
                report["loops"].append({"kernel": fname, "variant": hit, "instructions": hi - lo, "first": lo, "last": hi - 1})
        if sorted(found) != sorted(want):
            problems.append("%s: symbol loops found intact: %s, expected %s (s_getpc anchors: %d)" % (fname, found, want, len(anchors)))
        scratch = sum(1 for x in mn if x.startswith("scratch_"))
        report["kernels"].setdefault(fname, {})["scratch_instructions_outside_the_loops"] = scratch
    return report, problems


if __name__ == "__main__":
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "lzma_rs_amd", "libmilzma.so")
    rep, problems = check(so)
    for k, m in rep["kernels"].items():
        print("%-60s %s" % (k[:60], {a: b for a, b in m.items() if a != "name"}))
    for l in rep["loops"]:
        print("loop %-4s intact in %-50s %d instructions" % (l["variant"], l["kernel"][:50], l["instructions"]))
    for p in problems:
        print("PROBLEM: " + p)
    sys.exit(1 if problems else 0)

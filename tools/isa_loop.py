#!/usr/bin/env python
"""Static look at a kernel's candidate loop in the gfx950 ISA hipcc emits (no GPU needed):

    python tools/isa_loop.py render_bwd.hip k_render_bwd_tile            # summary: registers, scratch, LDS, per-loop instruction mix
    python tools/isa_loop.py render_fwd.hip 'k_render_fwdILb0' --dump    # ... and the innermost loop's instructions

Compiles luciddreamer_amd/csrc/<file> to assembly with the flags of luciddreamer_amd/build.py, finds the kernel whose mangled
name contains <pattern>, and for every `s_ff1_i32_b64` (the candidate walk of the blend kernels: next set bit of the mask)
counts the VALU / SALU / LDS / VMEM instructions and the `v_mov_b32` among them between it and the next one (or the end of the
kernel).  What round 4 used by hand to see whether a source change really removed an instruction -- or traded it for a copy."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from luciddreamer_amd import build  # noqa: E402


def main():
    src, pat = sys.argv[1], sys.argv[2]
    dump = "--dump" in sys.argv
    path = src if os.path.exists(src) else os.path.join(build.CSRC, src)
    flags = [f for f in build.COMMON_FLAGS if f != "-fPIC"] + build.SOURCES.get(os.path.basename(path), [])
    os.makedirs(build.OBJDIR, exist_ok=True)
    build._write_hash_header()
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        cmd = [build.hipcc()] + flags + ["-I", build.OBJDIR, "-S", "--cuda-device-only", "-o", out, path]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.exit(r.stderr[-2000:])
        text = open(out).read().splitlines()
    starts = [i for i, ln in enumerate(text) if re.match(r"^_Z\w+:", ln) and pat in ln]
    if not starts:
        names = sorted({ln.split(":")[0] for ln in text if re.match(r"^_Z\w+:", ln)})
        sys.exit("no kernel matches; kernels in this file:\n  " + "\n  ".join(names))
    for s in starts:
        e = next((i for i in range(s, len(text)) if text[i].startswith(".Lfunc_end")), len(text) - 1)    # (early exits end in s_endpgm too)
        meta = {}
        for ln in text[e:e + 120]:
            m = re.match(r";\s*(NumVgprs|ScratchSize|Occupancy|LDSByteSize|NumSgprs):\s*(\d+)", ln.strip())
            if m:
                meta.setdefault(m.group(1), m.group(2))
        body = [ln.strip() for ln in text[s:e + 1] if ln.strip() and not ln.strip().startswith(";")]
        print(f"{text[s].split(':')[0][:100]}\n  " + ", ".join(f"{k} {v}" for k, v in meta.items()))
        marks = [i for i, ln in enumerate(body) if ln.startswith("s_ff1_i32_b64")] + [len(body)]
        for a, b in zip(marks[:-1], marks[1:]):
            seg = body[a:b]
            kind = lambda p: sum(1 for ln in seg if ln.startswith(p))
            print(f"  walk at +{a}: {b - a} instructions to the next walk / end: VALU {kind('v_')} (v_mov {kind('v_mov_b32')}, "
                  f"v_cndmask {kind('v_cndmask')}), SALU {kind('s_') - kind('s_waitcnt') - kind('s_nop')}, LDS {kind('ds_')}, "
                  f"VMEM {kind('global_') + kind('buffer_') + kind('scratch_')}, branches {kind('s_cbranch') + kind('s_branch')}")
            if dump:
                print("\n".join("      " + ln for ln in seg))


if __name__ == "__main__":
    main()

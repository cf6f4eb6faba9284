#!/usr/bin/env python3
"""Per kernel of csrc/*.hip: number of global loads vs `s_waitcnt vmcnt(0)` / counted `vmcnt(N)` waits in the gfx950 ISA,
registers and spills.  A memory-bound kernel with about as many vmcnt(0) waits as loads has its loads serialised (one
memory round trip each): the usual cause is a load under a bounds test or a divergent branch around the load loop
(rule 1).  CPU only (hipcc cross-compiles):

    python tools/isa_waits.py [file.hip ...]
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rnnpose_amd import build  # noqa: E402

files = [os.path.abspath(f) for f in sys.argv[1:]] or sorted(glob.glob(os.path.join(ROOT, "rnnpose_amd", "csrc", "*.hip")))
for src in files:
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        flags = [f for f in build.FLAGS if f not in ("-shared", "-fPIC")] + build.PER_FILE_FLAGS.get(os.path.basename(src), [])
        cmd = [build.hipcc(), "-S", "--cuda-device-only", "-x", "hip", src, "-o", out, "-I", os.path.join(ROOT, "include"),
               "-I", os.path.join(ROOT, "rnnpose_amd", "csrc"), "-Rpass-analysis=kernel-resource-usage"] + flags
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=td)
        if not os.path.exists(out):
            print(src, "failed:\n", r.stderr[-2000:])
            continue
        res, name = {}, None
        for line in r.stderr.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                name = m.group(1)
                res[name] = {}
            for key in ("VGPRs:", "VGPRs Spill:", "Occupancy [waves/SIMD]:", "LDS Size [bytes/block]:"):
                if name and key in line:
                    res[name][key] = line.split(key)[1].split("[")[0].strip()
        txt = open(out).read()
        print(f"== {os.path.relpath(src, ROOT)}")
        for m in re.finditer(r"^(_Z\w+):.*?s_endpgm", txt, re.S | re.M):
            body, sym = m.group(0), m.group(1)
            dm = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
            short = re.sub(r"\(anonymous namespace\)::", "", dm).split("(")[0].replace("void ", "")[:58]
            loads = len(re.findall(r"global_load|buffer_load", body))
            w0 = len(re.findall(r"s_waitcnt vmcnt\(0\)", body))
            wn = len(re.findall(r"s_waitcnt vmcnt\([1-9]", body))
            ru = res.get(sym, {})
            print(f"  {short:58s} loads {loads:3d}  vmcnt(0) {w0:3d}  counted {wn:3d}  vgpr {ru.get('VGPRs:', '?'):>3s}  "
                  f"spill {ru.get('VGPRs Spill:', '?'):>3s}  waves/SIMD {ru.get('Occupancy [waves/SIMD]:', '?')}  "
                  f"lds {ru.get('LDS Size [bytes/block]:', '?')}")

"""Static check of csrc/conv3d_k3_c48.hip's compiled code: the fused epilogues load their operands with inline-asm loads the
compiler's waitcnt pass does not track (gload8_untracked) and complete them with one s_waitcnt vmcnt(16) placed after the
LDS-DMA pieces (wait_loads).  Correct only if NOTHING reads or writes the destination registers in between and no kernel
uses scratch.  usage: python tools/check_c48_isa.py   (compiles the file with -save-temps into a temporary directory)"""
import os
import re
import subprocess
import sys
import tempfile

here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(here, "medicalsemseg_amd", "csrc", "conv3d_k3_c48.hip")
with tempfile.TemporaryDirectory() as td:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-save-temps=obj", "-c", src,
                           "-o", os.path.join(td, "c48.o")], stderr=subprocess.DEVNULL)
    asm = open(os.path.join(td, "conv3d_k3_c48-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
bad = 0
for name, size in re.findall(r"\.name:\s+(\S*k3c48_kernel\S*)[\s\S]*?\.private_segment_fixed_size:\s+(\d+)", asm):
    if int(size):
        print("scratch in", name, size); bad += 1
lines = asm.split("\n")
i, seqs = 0, 0
while i < len(lines):
    if "global_load_dwordx2" in lines[i] and "ASMSTART" in lines[i - 1]:
        regs, j = set(), i
        while j < len(lines) and "vmcnt(16)" not in lines[j]:
            l = lines[j]
            m = re.search(r"global_load_dwordx2 v\[(\d+):(\d+)\]", l)
            if m and "ASMSTART" in lines[j - 1]:
                regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
            elif not l.strip().startswith(";") and "global_load_lds" not in l:
                used = set()
                for a, b in re.findall(r"v\[(\d+):(\d+)\]", l):
                    used.update(range(int(a), int(b) + 1))
                used.update(int(a) for a in re.findall(r"\bv(\d+)\b", l))
                if used & regs:
                    print("touches an in-flight register:", l.strip()); bad += 1
            j += 1
        assert j < len(lines), "load sequence without its wait"
        seqs += 1
        i = j
    i += 1
print(f"{seqs} load .. wait sequences, {bad} problems")
sys.exit(1 if bad or seqs == 0 else 0)

"""Where a kernel's scratch traffic sits: compile one HIP source to gfx950 assembly and list, for one kernel, every scratch load/store
with the kind of code around it (MFMA loop, LUT reads, global loads/stores), so that a spill in a cold path can be told from one in the
hot loop.  usage: python tools/spill_map.py seed_amd/csrc/gemm_bf16.hip _ZN12_GLOBAL__N_114gemm256_kernelILi2ELb1EEEvNS_10GemmParamsE"""
import subprocess
import sys

src, name = sys.argv[1], sys.argv[2]
asm = "/tmp/spill_map.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Iinclude", "-S", "--cuda-device-only", src, "-o", asm],
               check=True, stderr=subprocess.DEVNULL)
s = open(asm).read()
i = s.index("\n" + name + ":")
body = s[i:s.index("s_endpgm", i)].split("\n")
print(len(body), "lines")
last = None
for k, l in enumerate(body):
    t = None
    if "scratch_" in l:
        t = l.strip()[:72]
    elif "v_mfma" in l:
        t = "MFMA"
    elif "global_load_lds" in l:
        t = "DMA"
    elif "global_store" in l or "buffer_store" in l:
        t = "STORE"
    elif "global_load_dword" in l:
        t = "GLOAD"
    elif "ds_read_u16" in l:
        t = "LUT"
    if t and (t != last or "scratch" in t):
        print(k, t)
        last = t

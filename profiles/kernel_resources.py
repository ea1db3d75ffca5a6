"""Registers, spills and LDS of the kernels in the built library (the code object's metadata notes).
usage: python profiles/kernel_resources.py [substring ...]      (no GPU needed)"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = os.environ.get("RP_AMD_LIBRARY", os.path.join(ROOT, "commonroad-reactive-planner_amd", "lib", "librp_amd.so"))
with tempfile.TemporaryDirectory() as d:
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
    subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fat], check=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    f"--output={co}"], check=True)
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
ks = re.findall(r"\.group_segment_fixed_size: (\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size: (\d+).*?\.sgpr_count:\s+(\d+).*?\.sgpr_spill_count: (\d+)"
                r".*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count: (\d+)", notes, re.S)
names = subprocess.run(["c++filt"], input="\n".join(k[1] for k in ks), capture_output=True, text=True).stdout.splitlines()
want = sys.argv[1:]
for (lds, _, scratch, sg, ss, vg, vs), name in zip(ks, names):
    name = name.split("(")[0].replace("void ", "")
    if want and not any(w in name for w in want):
        continue
    print(f"{name:80s} vgpr {vg:>3s} (spilled {vs})  sgpr {sg:>3s} (spilled {ss})  lds {lds:>6s} B  scratch {scratch} B")

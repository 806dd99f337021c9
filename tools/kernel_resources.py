#!/usr/bin/env python3
"""Compiler-reported resources of the library's kernels (what rocprofv3's trace columns leave open: `VGPR_Count` there is
the ARCHITECTURAL registers only and `LDS_Block_Size` the STATIC allocation only -- the 2-D kernels size their LDS at
launch).  Reads the code objects inside csrc/_obj/*.o (objcopy .hip_fatbin -> clang-offload-bundler -> llvm-readelf
--notes) and prints one line per kernel:  python tools/kernel_resources.py [substring ...] > profiles/r04_kernel_resources.txt"""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
filters = sys.argv[1:]
rows = []
for obj in sorted(glob.glob(os.path.join(ROOT, "control_pcgrl_amd", "csrc", "_obj", "*.o"))):
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "co.elf")
        if subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat]).returncode or not os.path.getsize(fat):
            continue
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--input={fat}", f"--output={co}"], check=True)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    for blk in notes.split("  - .agpr_count:")[1:]:
        f = dict(re.findall(r"\.(\w+):\s+(\S+)", ".agpr_count:" + blk.split("\n  - .agpr_count")[0]))
        name = subprocess.run(["c++filt", f.get("name", "?")], capture_output=True, text=True).stdout.strip() or f.get("name", "?")
        name = name.split("(")[0].replace("void ", "")
        if filters and not any(s in name for s in filters):
            continue
        rows.append((os.path.basename(obj), name, int(f.get("vgpr_count", 0)), int(f.get("agpr_count", 0)), int(f.get("sgpr_count", 0)),
                     int(f.get("group_segment_fixed_size", 0)), int(f.get("private_segment_fixed_size", 0)), int(f.get("max_flat_workgroup_size", 0))))
print(f"{'object':26s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'static LDS':>10s} {'scratch':>8s} {'max wg':>6s}  kernel")
for r in rows:
    print(f"{r[0]:26s} {r[2]:5d} {r[3]:5d} {r[4]:5d} {r[5]:10d} {r[6]:8d} {r[7]:6d}  {r[1]}")

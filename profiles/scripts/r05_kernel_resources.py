"""Static resources of every hgx kernel in libhgx.so, from the code objects' AMDGPU metadata (llvm-readelf --notes): VGPRs, SGPRs,
scratch (private segment) and LDS bytes a workgroup, and the waves a SIMD can hold by the VGPR count (gfx950: 512 VGPRs a SIMD lane,
allocated in blocks of 8).  No GPU needed: what the compiler made of the kernels, not how they run.
usage: python profiles/scripts/r05_kernel_resources.py [libhgx.so] > profiles/r05_kernel_resources.txt"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "hal_amd", "libhgx.so")
    rows = {}
    with tempfile.TemporaryDirectory() as tmp:
        shutil.copyfile(lib, os.path.join(tmp, "lib.so"))
        subprocess.run([LLVM + "/llvm-objdump", "--offloading", "lib.so"], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for name in sorted(os.listdir(tmp)):
            if "amdgcn" not in name:
                continue
            notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", os.path.join(tmp, name)], stdout=subprocess.PIPE).stdout.decode()
            for block in notes.split("  - .agpr_count:")[1:]:
                def field(key, block=block):
                    m = re.search(r"\.%s:\s+(\S+)" % key, block)
                    return m.group(1) if m else "?"
                sym = field("name")
                dem = subprocess.run(["c++filt", sym], stdout=subprocess.PIPE).stdout.decode().strip()
                m = re.match(r"(?:void )?(?:hgx::)?(?:\(anonymous namespace\)::)?([A-Za-z_0-9]+)", dem)
                short = m.group(1) if m else dem
                if not short.startswith("k_"):
                    continue
                vg, sg = int(field("vgpr_count")), int(field("sgpr_count"))
                scratch, lds = int(field("private_segment_fixed_size")), int(field("group_segment_fixed_size"))
                waves = min(8, 512 // max(8, (vg + 7) // 8 * 8))
                key = (short, dem)
                rows[key] = (vg, sg, scratch, lds, waves, field("max_flat_workgroup_size"))
    print("%-34s %5s %5s %8s %7s %6s  %s" % ("kernel", "VGPR", "SGPR", "scratch", "LDS", "waves", "instantiation"))
    for (short, dem), (vg, sg, scratch, lds, waves, wg) in sorted(rows.items()):
        inst = dem[dem.index("<"):dem.index("(")] if "<" in dem and "(" in dem and dem.index("<") < dem.index("(") else ""
        print("%-34s %5d %5d %8d %7d %6d  %s" % (short, vg, sg, scratch, lds, waves, inst[:70]))


if __name__ == "__main__":
    main()

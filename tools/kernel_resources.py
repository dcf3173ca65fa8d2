"""Registers, LDS and scratch of every kernel in the built library, read from the code objects inside libldm_hip.so.

usage: kernel_resources.py [path/to/libldm_hip.so]      prints one line per kernel; exit status 1 when a kernel uses
scratch memory or spills registers (a kernel of this library that does has lost an array to private memory: the
fused up-sampling instantiation once went from 27.6 to 40.9 us that way without any test noticing)."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def kernel_resources(lib):
    """[{name, vgpr, sgpr, lds, scratch, spill}] for every kernel of every gfx950 code object in `lib`."""
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        for i, st in enumerate(starts):
            part = os.path.join(tmp, f"bundle{i}.bin")
            with open(part, "wb") as fh:
                fh.write(blob[st: starts[i + 1] if i + 1 < len(starts) else len(blob)])
            co = os.path.join(tmp, f"dev{i}.co")
            r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True, text=True)
            if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
            for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
                def field(key, default="0"):
                    m = re.search(r"\." + key + r":\s+(\S+)", blk)
                    return m.group(1) if m else default
                out.append({"name": field("name", "?"), "vgpr": int(field("vgpr_count")), "sgpr": int(field("sgpr_count")),
                            "lds": int(field("group_segment_fixed_size")), "scratch": int(field("private_segment_fixed_size")),
                            "spill": int(field("vgpr_spill_count")) + int(field("sgpr_spill_count"))})
    return out


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                              "diffusionmodel_amd", "libldm_hip.so")
    ks = kernel_resources(lib)
    bad = 0
    for k in sorted(ks, key=lambda k: k["name"]):
        flag = "  <-- scratch / spills" if k["scratch"] or k["spill"] else ""
        bad += bool(flag)
        print(f"{k['name'][:100]:100s} vgpr {k['vgpr']:4d} sgpr {k['sgpr']:4d} lds {k['lds']:6d} scratch {k['scratch']:5d} spill {k['spill']:3d}{flag}")
    print(f"{len(ks)} kernels, {bad} with scratch or spills")
    sys.exit(1 if bad else 0)

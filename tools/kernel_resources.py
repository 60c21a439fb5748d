#!/usr/bin/env python
"""Register / LDS / scratch budget of every device kernel in the built objects (affnet_amd/csrc/obj/*.o), read from the code
objects' own metadata - what the hardware is told, not what a comment says.

    python tools/kernel_resources.py                 -> markdown table (profiles/r04_kernel_resources.md is this output)
    python tools/kernel_resources.py --json          -> {kernel: {...}}

The metadata notes list a kernel's fields alphabetically: `.group_segment_fixed_size` stands BEFORE `.name` and belongs to the
kernel named after it, `.private_segment_fixed_size` / `.vgpr_count` / `.vgpr_spill_count` stand after `.name` (reading the
LDS size off the lines that follow a name gives the NEXT kernel's value).  This parser keeps the fields of one `- .agpr_count`
... block together.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "group_segment_fixed_size",
          "private_segment_fixed_size", "max_flat_workgroup_size")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def demangle(names):
    if not names:                 # (with no operands the demanglers would wait on stdin)
        return []
    for tool in ("/usr/bin/c++filt", os.path.join(LLVM, "llvm-cxxfilt")):
        try:
            out = subprocess.run([tool] + list(names), capture_output=True, text=True, check=True, stdin=subprocess.DEVNULL,
                                 timeout=60).stdout.strip().split("\n")
            if len(out) == len(names):
                return out
        except Exception:
            pass
    return list(names)


def object_kernels(obj):
    """[{name, field: value, ...}] of one host object with an embedded gfx950 code object."""
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
        # explicit output file: without one llvm-objcopy rewrites its INPUT in place (same bytes, new mtime - and build.sh compares mtimes)
        r = subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj, os.path.join(td, "copy.o")], capture_output=True)
        if r.returncode != 0 or not os.path.isfile(fat):      # a translation unit without device code (host glue only)
            return []
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=" + TARGET, "--input=" + fat,
                        "--output=" + co], check=True, capture_output=True)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
    kernels, cur = [], None
    in_kernels = False
    for line in notes.splitlines():
        if line.strip().startswith("amdhsa.kernels:"):
            in_kernels = True
            continue
        if not in_kernels:
            continue
        if re.match(r"^\s*amdhsa\.", line):          # next top-level key (amdhsa.target, amdhsa.version)
            break
        m = re.match(r"^(\s*)- \.(\w+):\s*(.*)$", line)
        if m and len(m.group(1)) <= 2:                # a new kernel block ("  - .agpr_count: 0")
            cur = {}
            kernels.append(cur)
            key, val = m.group(2), m.group(3)
        else:
            m = re.match(r"^\s+\.(\w+):\s*(.*)$", line)
            if not m or cur is None:
                continue
            key, val = m.group(1), m.group(2)
        if key == "name":
            cur["name"] = val.strip()
        elif key in FIELDS:
            cur[key] = int(val)
    return [k for k in kernels if "name" in k]


def all_kernels(obj_dir=None):
    obj_dir = obj_dir or os.path.join(ROOT, "affnet_amd", "csrc", "obj")
    out = {}
    for f in sorted(os.listdir(obj_dir)):
        if not f.endswith(".o"):
            continue
        ks = object_kernels(os.path.join(obj_dir, f))
        for k, dn in zip(ks, demangle([k["name"] for k in ks])):
            k["object"] = f
            k["demangled"] = re.sub(r"\(.*\)$", "", dn.replace("void ", "", 1))
            out[k["name"]] = k
    return out


def workgroups_per_cu(k):
    """Resident workgroups per CU by LDS (160 KB) and by registers (512 VGPRs per SIMD lane, waves of the workgroup spread over 4 SIMDs)."""
    lds = k.get("group_segment_fixed_size", 0)
    by_lds = 32 if lds == 0 else (160 * 1024) // lds
    waves = max(1, k.get("max_flat_workgroup_size", 256) // 64)
    regs = max(1, k.get("vgpr_count", 0) + k.get("agpr_count", 0))
    regs = (regs + 7) // 8 * 8
    waves_per_simd = min(8, 512 // regs)
    by_regs = (waves_per_simd * 4) // waves
    return max(0, min(by_lds, by_regs, 32))


# the kernels DESIGN.md section 4 quotes figures for (substring of the demangled name -> label)
DESIGN_KERNELS = [
    ("cnn32_trunk_kernel<2, 8, false, 0>", "HardNet trunk, exact fp32 MFMA"),
    ("cnn32_trunk_kernel<2, 8, false, 3>", "HardNet trunk, arith fp32_split3"),
    ("cnn32_trunk_kernel<2, 8, false, 2>", "HardNet trunk, arith fp32_split2h"),
    ("cnn32_trunk_kernel<0, 8, false, 0>", "AffNet trunk, exact fp32 MFMA"),
    ("cnn32_trunk_kernel<0, 8, false, 3>", "AffNet trunk, arith fp32_split3"),
    ("cnn32_trunk_kernel<0, 8, false, 2>", "AffNet trunk, arith fp32_split2h"),
    ("cnn32_trunk_kernel<1, 8, false, 0>", "OriNet trunk, exact fp32 MFMA"),
    ("cnn32_trunk_kernel<1, 8, false, 3>", "OriNet trunk, arith fp32_split3"),
    ("cnn32_trunk_kernel<1, 8, false, 2>", "OriNet trunk, arith fp32_split2h"),
    ("hardnet_head_kernel<64>", "HardNet head GEMM (64-patch tiles), exact"),
    ("hardnet_head_s3_kernel<64, 3>", "HardNet head GEMM (64-patch tiles), arith fp32_split3"),
    ("hardnet_head_s3_kernel<64, 2>", "HardNet head GEMM (64-patch tiles), arith fp32_split2h"),
    ("hessian_nms_kernel<5>", "Hessian + 3-D NMS + centroid, 5 levels per octave"),
    ("blur2d_kernel<13, 4>", "Gaussian blur 13 x 13, 64 x 64 tiles"),
    ("blur2d_pair_kernel<15, 9, 4>", "paired blur 15 x 15 / 9 x 9"),
    ("dense_conv_kernel<32, 32, 1", "dense AffNetFastFullConv conv3, exact"),
    ("dense_conv_s3_kernel<32, 32, 1, LayQ<16, 16, 18, 32, 0, 3>", "dense AffNetFastFullConv conv3, arith fp32_split3"),
    ("dense_conv_s3_kernel<32, 32, 1, LayR<16, 16, 20, 32", "dense AffNetFastFullConv conv3, arith fp32_split2h"),
]


def design_table(ks=None):
    ks = ks or all_kernels()
    lines = ["| kernel | role | VGPR | LDS bytes | scratch bytes | VGPR spills | workgroups / CU |", "|---|---|---|---|---|---|---|"]
    for pat, label in DESIGN_KERNELS:
        hit = [k for k in ks.values() if pat in k["demangled"]]
        if not hit:
            lines.append("| `%s` | %s | (not built) | | | | |" % (pat, label))
            continue
        k = hit[0]
        lines.append("| `%s` | %s | %d | %d | %d | %d | %d |" % (k["demangled"], label, k.get("vgpr_count", 0) + k.get("agpr_count", 0), k.get("group_segment_fixed_size", 0),
                                                             k.get("private_segment_fixed_size", 0), k.get("vgpr_spill_count", 0), workgroups_per_cu(k)))
    return "\n".join(lines)


def main():
    ks = all_kernels()
    if "--json" in sys.argv:
        print(json.dumps(ks, indent=1, sort_keys=True))
        return
    if "--design" in sys.argv:
        print(design_table(ks))
        return
    print("| kernel | object | VGPR (+AGPR) | SGPR | LDS bytes | scratch bytes | VGPR spills | workgroups / CU (LDS, registers) |")
    print("|---|---|---|---|---|---|---|---|")
    for name in sorted(ks, key=lambda n: (ks[n]["object"], ks[n]["demangled"])):
        k = ks[name]
        print("| `%s` | %s | %d (+%d) | %d | %d | %d | %d | %d |" % (k["demangled"], k["object"], k.get("vgpr_count", 0), k.get("agpr_count", 0), k.get("sgpr_count", 0),
                                                              k.get("group_segment_fixed_size", 0), k.get("private_segment_fixed_size", 0),
                                                              k.get("vgpr_spill_count", 0), workgroups_per_cu(k)))


if __name__ == "__main__":
    main()

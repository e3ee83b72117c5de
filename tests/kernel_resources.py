"""Register / scratch / LDS figures of the kernels inside the built libpetal_hip.so, read from the code object's notes.

The library's `.hip_fatbin` section is an offload bundle; its gfx950 member is an ELF whose AMDGPU metadata note lists, per
kernel, the VGPR / AGPR / SGPR counts, spills, scratch bytes and static LDS.  No GPU needed (`tests/test_kernel_budgets.py`
asserts budgets for the hot instantiations; `python tests/kernel_resources.py [regex]` prints the table).
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("PETAL_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
LIB = os.path.join(ROOT, "petal-decomposition_amd", "libpetal_hip.so")
FIELDS = {
    ".vgpr_count": "vgpr", ".agpr_count": "agpr", ".sgpr_count": "sgpr", ".vgpr_spill_count": "vgpr_spill",
    ".sgpr_spill_count": "sgpr_spill", ".private_segment_fixed_size": "scratch", ".group_segment_fixed_size": "lds",
    ".max_flat_workgroup_size": "max_wg",
}


def _run(*cmd):
    return subprocess.run(cmd, check=True, capture_output=True, text=True).stdout


def kernel_resources(lib: str = LIB) -> dict:
    """{demangled kernel name: {vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, scratch, lds, max_wg, waves_per_simd}}"""
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        _run(os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", lib, os.path.join(tmp, "copy.so"))
        _run(os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
             "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}")
        notes = _run(os.path.join(LLVM, "llvm-readelf"), "--notes", co)
    kernels, cur = {}, None
    entries = []
    for line in notes.splitlines():
        m = re.match(r"^(  - |    )(\.[a-z_]+):\s*(.*)$", line)   # kernel-level keys only (argument keys sit deeper)
        if not m:
            continue
        if m.group(1) == "  - ":
            cur = {}
            entries.append(cur)
        if cur is None:
            continue
        key, val = m.group(2), m.group(3).strip()
        if key == ".name":
            cur["mangled"] = val.strip("'")
        elif key in FIELDS:
            cur[FIELDS[key]] = int(val)
    names = [e["mangled"] for e in entries if "mangled" in e and "vgpr" in e]
    dem = subprocess.run(["c++filt"], input="\n".join(names), check=True, capture_output=True, text=True).stdout.splitlines()
    for e, name in zip([e for e in entries if "mangled" in e and "vgpr" in e], dem):
        e = dict(e)
        e.pop("mangled")
        # gfx950: 512 unified registers per SIMD lane, allocated in blocks of 8 (VGPR + AGPR together)
        regs = e["vgpr"] + 0  # .vgpr_count already holds the unified total (VGPRs + AGPRs) on gfx90a+
        blocks = max(1, -(-regs // 8) * 8)
        e["waves_per_simd"] = min(8, 512 // blocks)
        kernels[re.sub(r"\(.*$", "", name)] = e
    return kernels


if __name__ == "__main__":
    pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else ".")
    res = kernel_resources(sys.argv[2] if len(sys.argv) > 2 else LIB)
    print(f"{'kernel':70s} vgpr agpr sgpr spill scratch   lds waves/SIMD")
    for name, r in sorted(res.items()):
        if pat.search(name):
            print(f"{name[:70]:70s} {r['vgpr']:4d} {r.get('agpr', 0):4d} {r['sgpr']:4d} {r['vgpr_spill']:5d} {r['scratch']:7d} "
                  f"{r.get('lds', 0):5d} {r['waves_per_simd']:3d}")

#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch figures of the SHIPPED library, read from the code object's own metadata
(the AMDGPU notes of the gfx950 ELF inside the .so's .hip_fatbin bundle) — not from rocprofv3's kernel records, whose
VGPR field saturates at 128 on this stack.

  python tools/kernel_resources.py [lib.so] [--all]     ->  one line per kernel (demangled name)

waves/SIMD = min(floor(512 / alloc(unified registers)), 8, LDS limit); `regs` = unified count (of which AGPR) with the gfx950 figures of
/opt/skills/guides/MI355X_MICROARCH.md: 512 registers per lane and SIMD, allocation granule 8, 160 KB LDS per CU shared by
the workgroups of its four SIMDs (every kernel here is one wavefront per workgroup, except assemble_kernel: 256 lanes).
"""
import re, struct, subprocess, sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
CXXFILT = "c++filt"


def gfx950_code_object(so: Path) -> bytes:
    data = so.read_bytes()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    at = data.find(magic)
    if at < 0:
        raise SystemExit(f"{so}: no offload bundle")
    p = at + len(magic)
    (n,) = struct.unpack_from("<Q", data, p)
    p += 8
    for _ in range(n):
        off, size, idlen = struct.unpack_from("<QQQ", data, p)
        p += 24
        ident = data[p : p + idlen].decode()
        p += idlen
        if "gfx950" in ident:
            return data[at + off : at + off + size]
    raise SystemExit(f"{so}: no gfx950 entry in the bundle")


def kernels(so: Path):
    tmp = Path("/tmp") / (so.stem + ".gfx950.co")
    tmp.write_bytes(gfx950_code_object(so))
    notes = subprocess.run([READELF, "--notes", str(tmp)], capture_output=True, text=True, check=True).stdout
    out = []
    for block in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
        block = ".agpr_count:" + block
        f = {k: v.strip() for k, v in re.findall(r"\.(\w+):\s*([^\n]+)", block)}
        if "name" not in f or "vgpr_count" not in f:
            continue
        out.append(f)
    names = subprocess.run([CXXFILT], input="\n".join(k["name"] for k in out), capture_output=True, text=True).stdout.split("\n")
    for k, nm in zip(out, names):
        nm = re.sub(r"\(.*", "", nm).replace("void ", "").replace("fsdp::", "")
        k["short"] = nm
    return out


def occupancy(vgpr, agpr, lds, wg_lanes):
    alloc = -(-max(vgpr, 1) // 8) * 8  # (.vgpr_count of a gfx90a+ code object is the unified total: architectural + accumulation registers)
    by_regs = min(8, 512 // alloc)
    waves_per_wg = max(1, wg_lanes // 64)
    if lds:
        wgs_per_cu = (160 * 1024) // lds
        by_lds = wgs_per_cu * waves_per_wg / 4.0
    else:
        by_lds = 8
    return by_regs, by_lds


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    so = Path(args[0]) if args else ROOT / "ft-fsd-path-planning_amd" / "lib" / "libfsdp_hip.so"
    print(f"# {so.name}: kernel resources from the code object metadata (llvm-readelf --notes of the gfx950 ELF in .hip_fatbin)")
    print(f"{'kernel':44s} {'regs':>5s} {'AGPR':>5s} {'SGPR':>5s} {'VGPR spill':>10s} {'SGPR spill':>10s} {'scratch B/lane':>14s} {'LDS B/wg':>9s} {'waves/SIMD (regs | LDS)':>24s}")
    for k in sorted(kernels(so), key=lambda k: k["short"]):
        v, a, s = int(k["vgpr_count"]), int(k.get("agpr_count", 0)), int(k["sgpr_count"])
        lds, scr = int(k["group_segment_fixed_size"]), int(k["private_segment_fixed_size"])
        wg = int(k.get("max_flat_workgroup_size", 64))
        regs, by_lds = occupancy(v, a, lds, 256 if "assemble" in k["short"] else 64)
        print(f"{k['short'][:44]:44s} {v:5d} {a:5d} {s:5d} {int(k.get('vgpr_spill_count', 0)):10d} {int(k.get('sgpr_spill_count', 0)):10d} {scr:14d} {lds:9d} {regs:10d} | {by_lds:5.2f}")


if __name__ == "__main__":
    main()

"""Build-time check for the L2-warming variant of the narrow-layer kernel (isf_spconv_dma.hip, spconv_dma_warm_kernel).

That variant issues loads from inline asm into three NAMED registers at the top of its 96-register budget (v93, v94:
table words in flight; v95: destination of the throw-away loads) and leaves them in flight across its per-step wait.
The asm clobber lists keep the compiler from holding a value in them ACROSS those statements, but nothing stops it from
using them in between -- a load still in flight would then land on the compiler's value.  The compiler allocates from
v0 upwards, so it only gets there under more register pressure than the kernels have today; this script makes that a
build failure instead of a wrong result: it compiles the device code to assembly and fails if, in any warming kernel,
an instruction the compiler generated (anything outside the #ASMSTART/#ASMEND brackets) names v93..v95, alone or
inside a register range, or if the kernel uses accumulation registers or scratch (whose offsets / reloads the scheme
does not account for), or needs more than 96 registers.

    python tools/check_reserved_regs.py [path/to/isf_spconv_dma.hip]     -> exit code 0 / 1
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "..", "is-fusion_amd", "csrc", "isf_spconv_dma.hip")
RESERVED = {93, 94, 95}
ONE = re.compile(r"(?<![\w.])v(\d+)\b")
RANGE = re.compile(r"(?<![\w.])v\[(\d+):(\d+)\]")


def device_asm(src):
    inc = os.path.join(HERE, "..", "include")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-I", inc,
           "-o", "-", src]
    return subprocess.run(cmd, check=True, capture_output=True, text=True).stdout


def warming_kernels(asm):
    """{kernel name: ([(line number, text)] of compiler-generated instructions, {registers the asm statements name})}
    for the warming kernels."""
    out, name, in_asm = {}, None, False
    for n, line in enumerate(asm.splitlines(), 1):
        m = re.match(r"^(_ZN3isf22spconv_dma_warm_kernel\w*):", line)
        if m:
            name = m.group(1)
            out[name] = ([], set())
            continue
        if re.match(r"^\w+:", line):
            name = None
            continue
        if name is None:
            continue
        if "s_endpgm" in line:
            name = None
            continue
        if "#ASMSTART" in line:
            in_asm = True
        elif "#ASMEND" in line:
            in_asm = False
        elif in_asm:
            out[name][1].update(int(r) for r in ONE.findall(line.split(";")[0]))
        else:
            code = line.split(";")[0]
            if code.strip() and not code.strip().startswith("."):
                out[name][0].append((n, code))
    return out


def offenders(kernels):
    bad = []
    for name, (lines, reserved) in kernels.items():
        if not RESERVED <= reserved:
            bad.append((name, 0, f"the asm statements name {sorted(reserved & RESERVED)}: expected v93, v94, v95"))
        reserved = RESERVED
        for n, code in lines:
            hit = any(int(r) in reserved for r in ONE.findall(code))
            hit = hit or any(any(lo <= r <= hi for r in reserved) for lo, hi in ((int(a), int(b)) for a, b in RANGE.findall(code)))
            if hit:
                bad.append((name, n, code.strip()))
    return bad


def resources(asm):
    """[(kernel, problem)] from the kernel descriptors' metadata: accumulation registers, scratch, > 96 registers."""
    bad = []
    meta = asm[asm.find("amdhsa.kernels:"):]
    for item in re.split(r"\n  - (?=\.)", meta)[1:]:
        m = re.search(r"\.name:\s+(_ZN3isf22spconv_dma_warm_kernel\w*)", item)
        if not m:
            continue
        for key, val in re.findall(r"\.(agpr_count|private_segment_fixed_size|vgpr_count|vgpr_spill_count):\s+(\d+)", item):
            if (key == "vgpr_count" and int(val) > 96) or (key != "vgpr_count" and int(val) != 0):
                bad.append((m.group(1), f"{key} = {val}"))
    return bad


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else SRC
    asm = device_asm(src)
    kernels = warming_kernels(asm)
    for name, what in resources(asm):
        print(f"{name}: {what}")
        kernels.setdefault("!" + name, ([(0, "v93")], set(RESERVED)))
    if not kernels:
        print("no warming kernels found in", src)
        return 1
    bad = offenders(kernels)
    for name, n, code in bad[:20]:
        print(f"{name}: line {n}: {code}")
    print(f"{len(kernels)} warming kernels, {len(bad)} violations of the reserved registers v93..v95")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

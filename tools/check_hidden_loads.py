"""Static proof obligation of isf_spconv_ring.hip: the activation gathers of the ring kernel are `global_load_dwordx4`
instructions inside inline asm ("hidden" from hipcc's wait-count bookkeeping, so that the kernel can keep several stages
of loads in flight behind ONE counted `s_waitcnt vmcnt(N)` per step).  The hardware does not interlock a VGPR that a
load has not written yet, so this is only correct if, in the generated code,

  (1) between a hidden load and the `; HIDDEN_LANDED` marker that follows the covering `s_waitcnt`, NO instruction reads
      or writes a destination register of that load -- no compiler-inserted copy, spill, reuse or early MFMA (a forward
      "may be in flight" dataflow over the kernel's basic blocks to a fixed point: every path, back edges included);
  (2) the kernel does not spill (scratch) at all.

    python tools/check_hidden_loads.py            # compiles is-fusion_amd/csrc/isf_spconv_ring.hip to ISA and checks it
    python tools/check_hidden_loads.py file.s     # checks an existing ISA listing

Run by __graft_entry__.build(); exit code 1 and a list of violations otherwise."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def check_function(name, lines):
    """lines: the listing of one kernel -> list of violation strings.  Forward may-be-in-flight dataflow over the
    kernel's basic blocks (hipcc moves cold MFMA blocks out of line, so text order is not execution order)."""
    blocks, labels = [[]], {}
    for raw in lines:
        s = raw.strip()
        if not s:
            continue
        if re.match(r"^[.\w$]+:\s*(;.*)?$", s):                          # label

            if blocks[-1]:
                blocks.append([])
            labels[s.split(":")[0]] = len(blocks) - 1
            continue
        if s.startswith((".", "//")):                                   # directive
            continue
        code, _, comment = s.partition(";")
        code = code.strip()
        if "HIDDEN_LANDED" in comment or (not code and "HIDDEN_LANDED" in s):
            blocks[-1].append(("landed", s, regs_of(s.split("HIDDEN_LANDED", 1)[1]), None))
            continue
        if not code:
            continue
        mn = code.split(None, 1)[0]
        if "HIDDEN_LOAD" in comment:
            blocks[-1].append(("load", s, regs_of(code.split(None, 1)[1].split(",")[0]), None))
        elif "HIDDEN_WAIT" in comment:
            blocks[-1].append(("wait0" if re.search(r"vmcnt\(0\)", code) else "wait", s, set(), None))
        elif mn in ("s_branch", "s_endpgm") or mn.startswith("s_cbranch"):
            blocks[-1].append((mn, s, set(), code.split()[-1] if mn != "s_endpgm" else None))
            blocks.append([])
        else:
            blocks[-1].append(("inst", s, regs_of(code), None))
    if not any(t[0] == "load" for b in blocks for t in b):
        return [f"{name}: no hidden loads found (marker missing?)"]
    succ = []
    for i, b in enumerate(blocks):
        last = b[-1] if b else None
        if last and last[0] == "s_endpgm":
            succ.append([])
        elif last and last[0] == "s_branch":
            succ.append([labels[last[3]]])
        elif last and last[0].startswith("s_cbranch"):
            succ.append([labels[last[3]]] + ([i + 1] if i + 1 < len(blocks) else []))
        else:
            succ.append([i + 1] if i + 1 < len(blocks) else [])
    IN = [set() for _ in blocks]
    bad = set()
    work = list(range(len(blocks)))
    while work:
        i = work.pop()
        cur = set(IN[i])
        for kind, s, regs, _ in blocks[i]:
            if kind == "load":
                if regs & cur:
                    bad.add(f"{name}: hidden load into registers that may still be in flight {sorted(regs & cur)[:4]}: {s}")
                cur |= regs
            elif kind == "landed":
                cur -= regs
            elif kind == "wait0":
                cur = set()
            elif kind == "inst" and regs & cur:
                bad.add(f"{name}: instruction touches registers that may be in flight {sorted(regs & cur)[:4]}: {s}")
        for j in succ[i]:
            if not cur <= IN[j]:
                IN[j] |= cur
                work.append(j)
    return sorted(bad)


def check_listing(text):
    bad, n = [], 0
    cur, name = None, None
    for line in text.splitlines():
        m = re.match(r"^(_ZN3isf18spconv_ring_kernel\w+):", line)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            cur.append(line)
            if line.startswith(".Lfunc_end"):
                bad += check_function(name, cur)
                n += 1
                cur = None
    return bad, n


def main():
    if len(sys.argv) > 1:
        text = open(sys.argv[1]).read()
    else:
        src = os.path.join(ROOT, "is-fusion_amd", "csrc", "isf_spconv_ring.hip")
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "ring.s")
            r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-munsafe-fp-atomics",
                                "-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(src), "-S", "--cuda-device-only",
                                "-Rpass-analysis=kernel-resource-usage", "-o", out, src], capture_output=True, text=True)
            if r.returncode != 0:
                print(r.stderr[-3000:])
                return 1
            text = open(out).read()
            spills = [ln for ln in re.split(r"remark: Function Name: ", r.stderr)[1:]
                      if "spconv_ring_kernel" in ln and not re.search(r"ScratchSize \[bytes/lane\]: 0\b", ln)]
            if spills:
                print("kernels with scratch (spills) -- hidden loads cannot be proven safe:")
                for s in spills:
                    print("  ", s.split()[0])
                return 1
    bad, n = check_listing(text)
    for b in bad[:40]:
        print(b)
    print(f"check_hidden_loads: {n} ring kernels checked, {len(bad)} violations")
    return 1 if bad or n == 0 else 0


if __name__ == "__main__":
    sys.exit(main())

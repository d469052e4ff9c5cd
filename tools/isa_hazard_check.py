#!/usr/bin/env python3
"""Static gate for the gfx950 wide-store data hazard (tools/isa_hazard_check.py [LIB.so ...]).

Measured on MI355X (tools/store_hazard_repro.hip, profiles/r03_store_hazard_repro.txt): a vector-memory store of MORE
than 64 bits reads its data VGPRs over several cycles after issue, and a VALU instruction that overwrites one of them
too early corrupts lanes 12..15 of every 16-lane row of the stored data -- rarely (~0.015 % of lanes under load), and
differently from run to run.  Wait states needed between the store and the overwriting instruction, as measured:

    buffer_store_dwordx3/x4 with the soffset in an SGPR      1     (LLVM / ROCm 7.2 assumes 0: "no hazard")
    buffer_store_dwordx4 with a literal soffset              2     (LLVM inserts 1)
    global_store_dwordx4 (saddr form)                        2     (LLVM inserts 1)
    64-bit stores                                            0

So hipcc's output can be wrong whenever the register allocator happens to reuse a data register of a wide store
in the next one or two instructions -- which is what made the audited two-frame instantiation of k_mog_fused
update the model wrongly in round 2.  This script disassembles every gfx950 code object of the given libraries
(default: oat_amd/lib/liboatgpu.so) and fails if any >64-bit VMEM store is followed, within REQUIRED = 2 wait
states on any path (fall-through and branch targets), by an instruction that writes one of its data VGPRs.
It runs in the CPU test suite (tests/test_isa_hazards.py) and from `make`: the shipped binary is checked, not the
compiler trusted."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

def _find_objdump():
    """llvm-objdump of the ROCm in use: next to hipcc's clang (hipcc --print-prog-name), under $ROCM_PATH, /opt/rocm, PATH."""
    cands = []
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    try:
        r = subprocess.run([hipcc, "--print-prog-name=llvm-objdump"], capture_output=True, text=True, timeout=60)
        if r.returncode == 0 and r.stdout.strip():
            cands.append(r.stdout.strip().splitlines()[-1])
    except (OSError, subprocess.SubprocessError):
        pass
    for root in (os.environ.get("ROCM_PATH"), "/opt/rocm"):
        if root:
            cands.append(os.path.join(root, "lib", "llvm", "bin", "llvm-objdump"))
    cands.append(shutil.which("llvm-objdump") or "")
    for c in cands:
        if c and os.path.isabs(c) and os.path.exists(c):
            return c
    return "/opt/rocm/lib/llvm/bin/llvm-objdump"


OBJDUMP = _find_objdump()
REQUIRED = 2          # wait states wanted behind every wide store before a data VGPR may be written
WIDE = re.compile(r"^(buffer|global|flat|scratch)_store_(dwordx3|dwordx4|b96|b128)\b|^buffer_store_format_xyzw?\b|^tbuffer_store_format_xyzw?\b")
VREG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")      # v = VGPR n, a = AGPR n (kept apart as 1000 + n)
# instructions that never write a VGPR (first operand is not a VGPR destination)
NO_VDST = re.compile(r"^(s_|buffer_store|global_store|flat_store|scratch_store|ds_write|ds_store|tbuffer_store|v_cmp|v_cmpx|"
                     r"global_atomic_\w+ (?!v)|buffer_atomic|v_nop|v_readlane|v_readfirstlane|buffer_wbl2|buffer_inv|buffer_gl)")


def code_objects(lib, tmp):
    """Extract the gfx950 code objects of a HIP shared library into tmp (llvm-objdump --offloading writes them next
    to its input, so the library is copied first)."""
    local = os.path.join(tmp, os.path.basename(lib))
    shutil.copy(lib, local)
    subprocess.run([OBJDUMP, "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if f.startswith(os.path.basename(lib) + ".") and "amdgcn" in f)


def parse(co):
    """-> {function: [(address, mnemonic + operands)]}"""
    out = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
    funcs, cur = {}, None
    for line in out.splitlines():
        m = re.match(r"^([0-9a-f]+) <([^>]+)>:", line)
        if m:
            cur = funcs.setdefault(m.group(2), [])
            continue
        m = re.match(r"^\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if m and cur is not None:
            cur.append((int(m.group(2), 16), m.group(1).strip()))
    return funcs


def regs(text):
    r = set()
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            r.add(int(m.group(2)) + (1000 if m.group(1) == "a" else 0))
        else:
            off = 1000 if m.group(3) == "a" else 0
            r.update(range(int(m.group(4)) + off, int(m.group(5)) + 1 + off))
    return r


def store_data_regs(ins):
    ops = ins.split(None, 1)[1]
    parts = [p.strip() for p in ops.split(",")]
    # buffer_store: vdata first; global/flat/scratch_store: address first, data second
    data = parts[0] if ins.startswith(("buffer_", "tbuffer_")) else parts[1]
    return regs(data)


def written_vgprs(ins):
    if NO_VDST.match(ins):
        return set()
    ops = ins.split(None, 1)
    if len(ops) < 2:
        return set()
    first = ops[1].split(",")[0]
    w = regs(first)
    if ins.startswith("v_swap_b32"):
        w |= regs(ops[1].split(",")[1])
    return w


def wait_states(ins):
    m = re.match(r"^s_nop (\d+)", ins)
    return int(m.group(1)) + 1 if m else 1


def check_function(name, body):
    index = {a: i for i, (a, _) in enumerate(body)}
    bad = []
    for i, (addr, ins) in enumerate(body):
        if not WIDE.match(ins):
            continue
        data = store_data_regs(ins)
        # walk every path for REQUIRED wait states
        work = [(i + 1, 0)]
        seen = set()
        while work:
            j, ws = work.pop()
            if ws >= REQUIRED or j >= len(body) or (j, ws) in seen:
                continue
            seen.add((j, ws))
            a2, nxt = body[j]
            hit = written_vgprs(nxt) & data
            if hit:
                bad.append((name, addr, ins, a2, nxt, ws, sorted(hit)))
                continue
            if nxt.startswith("s_endpgm"):
                continue
            if nxt.startswith(("s_setpc", "s_swappc")):
                # an indirect jump inside the window cannot be followed: count it as a hazard (the kernels of this library
                # are fully inlined; none has one)
                bad.append((name, addr, ins, a2, nxt, ws, ["indirect jump within the hazard window"]))
                continue
            m = re.match(r"^s_c?branch\w*\s+(\d+)", nxt)
            if m:                                   # objdump prints the relative simm16 as an unsigned number
                rel = int(m.group(1))
                rel -= 0x10000 if rel >= 0x8000 else 0
                tgt = a2 + 4 + 4 * rel
                if tgt in index:
                    work.append((index[tgt], ws + 1))
                if nxt.startswith("s_branch"):
                    continue
            work.append((j + 1, ws + wait_states(nxt)))
    return bad


def check(libs):
    bad, n_stores, n_funcs = [], 0, 0
    with tempfile.TemporaryDirectory() as tmp:
        for lib in libs:
            for co in code_objects(lib, tmp):
                for name, body in parse(co).items():
                    n_funcs += 1
                    n_stores += sum(1 for _, ins in body if WIDE.match(ins))
                    bad += check_function(name, body)
    return bad, n_stores, n_funcs


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libs = sys.argv[1:] or [os.path.join(here, "oat_amd", "lib", "liboatgpu.so")]
    if not os.path.exists(OBJDUMP):
        # an unchecked binary must not ship by accident: fail unless the builder says so explicitly
        if os.environ.get("OATGPU_SKIP_HAZARD_CHECK") == "1":
            print(f"isa_hazard_check: {OBJDUMP} not found and OATGPU_SKIP_HAZARD_CHECK=1 -- binary NOT checked")
            sys.exit(0)
        print(f"isa_hazard_check: llvm-objdump not found ({OBJDUMP}; tried hipcc --print-prog-name, $ROCM_PATH, /opt/rocm, PATH): the "
              "binary cannot be checked for the gfx950 wide-store hazard.  Set OATGPU_SKIP_HAZARD_CHECK=1 to build without the gate.")
        sys.exit(2)
    bad, n_stores, n_funcs = check(libs)
    for name, addr, ins, a2, nxt, ws, hit in bad:
        print(f"HAZARD {name}: {addr:#x} `{ins}` then after {ws} wait state(s) {a2:#x} `{nxt}` writes {hit} (1000 + n = AGPR n)")
    print(f"{len(bad)} wide-store hazards in {n_funcs} functions / {n_stores} stores of more than 64 bits ({', '.join(os.path.basename(l) for l in libs)})")
    sys.exit(1 if bad else 0)

"""ISA-level check of the wait states behind MFMA results (runs here, no GPU).  hipcc's hazard recogniser pads the MFMAs it emits itself; it
does NOT look inside `asm volatile` statements (csrc/oard_node_v1.h: dense_seq_xyz, csrc/oard_wgrad_t16.h), where the code carries its own
s_nops.  This walks the disassembly of a device code object: behind every v_mfma it counts issue states (an instruction = 1, `s_nop N` = N + 1)
and reports any instruction other than an MFMA accumulating IN PLACE on the same registers (vdst == srcC: the back-to-back chain the hardware
forwards) that reads or writes a register of the MFMA's destination before the result may be touched.
Required: passes + 2 states BETWEEN an MFMA and a non-MFMA instruction touching its destination (the toucher then issues in state
passes + 3: LLVM's GFX940_XDL_N_PassWriteVgprVALU{Raw,Waw}WaitStates 5 / 7 / 11 / 19 for 2 / 4 / 8 / 16 passes).  Calibration: with this number
hipcc's OWN code (whose pads its hazard recogniser computed) has no violation in any kernel of the library and sits exactly AT the
limit in hundreds of places; one state more and it is flagged 186 times.  An MFMA that takes the destination whole as its SrcC is the
accumulate chain (0 states), whatever its vdst; an MFMA that overwrites it is ordered by the in-order XDL pipe.
usage: python tools/mfma_hazard_check.py <device.o> [kernel-name regex]"""
import re
import subprocess
import sys

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
REG = re.compile(r"(?<![A-Za-z0-9_])([va])(?:\[(\d+):(\d+)\]|(\d+)\b)")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        bank = m.group(1)
        if m.group(4) is not None:
            out.add((bank, int(m.group(4))))
        else:
            out.update((bank, r) for r in range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def passes(op):
    """Issue passes (4 cycles each) of an MFMA opcode on gfx950."""
    if "_4x4x" in op:
        return 2
    if "f32_16x16x4_f32" in op or "32x32x2_f32" in op or "f64" in op:
        return 8 if "16x16" in op else 16
    if "32x32" in op:
        return 8 if "x16" in op or "x8" in op else 16
    return 4 if ("x32" in op or "x16" in op) else 8


BRANCHES = ("s_endpgm", "s_branch", "s_cbranch_scc0", "s_cbranch_scc1", "s_cbranch_vccz", "s_cbranch_vccnz", "s_cbranch_execz",
            "s_cbranch_execnz", "s_setpc_b64")


def is_mfma(op):
    return op.startswith(("v_mfma", "v_smfmac"))


def scan(code, name=""):
    """code: instruction strings of one function.  -> (violations, MFMAs seen)"""
    bad, n_mfma = [], 0
    for i, ins in enumerate(code):
        op = ins.split()[0]
        if not is_mfma(op):
            continue
        n_mfma += 1
        dst = regs(ins[len(op):].split(",")[0])
        need = passes(op) + 2                   # instructions / nop states BETWEEN the MFMA and the toucher (the toucher is state passes + 3)
        states = 0
        for nxt in code[i + 1:]:
            if states >= need:
                break
            nop = nxt.split()[0]
            if nop in BRANCHES:
                break                           # control flow: the linear scan ends (loops re-enter code the scan has covered)
            if nop == "s_nop":
                states += int(nxt.split()[1], 0) + 1
                continue
            if regs(nxt[len(nop):]) & dst:
                nops = [o.strip() for o in nxt[len(nop):].split(",")]
                srcc = nops[3].split()[0] if len(nops) > 3 else ""
                # an MFMA behind an MFMA: the XDL pipe is in order - only READING the result as SrcA / SrcB, or as a SrcC that is not
                # exactly the destination, waits; taking it whole as SrcC (the accumulate chain) and overwriting it do not
                chain = is_mfma(nop) and not (regs(",".join(nops[1:3])) & dst) and (regs(srcc) == dst or not (regs(srcc) & dst))
                if not chain:
                    bad.append(f"{name[:80]}: `{ins}` then after {states} states `{nxt}` (needs {need})")
                break                           # a chain's next link takes over: its own window is checked when the scan reaches it
            states += 1
    return bad, n_mfma


def disassemble(obj):
    dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", obj], capture_output=True, text=True, check=True).stdout
    kernels, cur = {}, None
    for ln in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
        if m:
            cur = m.group(1)
            kernels[cur] = []
            continue
        if cur is None or not ln[:1].isspace():
            continue
        ins = ln.split("//")[0].strip()
        if ins:
            kernels[cur].append(ins)
    return kernels


def check(obj, pattern=None):
    kernels = disassemble(obj)
    bad, n_mfma = [], 0
    for name, code in kernels.items():
        if pattern and not re.search(pattern, name):
            continue
        b, n = scan(code, name)
        bad += b
        n_mfma += n
    return bad, n_mfma, len(kernels)


if __name__ == "__main__":
    bad, n, k = check(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
    print(f"{k} functions, {n} MFMAs checked, {len(bad)} violations")
    for b in bad[:40]:
        print("  " + b)
    sys.exit(1 if bad else 0)

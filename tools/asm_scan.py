"""Static scan of the gfx950 assembly of every kernel in pointasnl_amd/csrc (no GPU needed: hipcc cross-compiles).

  python tools/asm_scan.py [--out profiles/<tag>_asm_scan.txt] [file.hip ...]

Per kernel: registers / scratch, and the two patterns DESIGN.md 6 describes ("waits across a loop's back edge", "loads behind
a select"):
  * predicated loads   -- `s_cbranch_execz` ... ONE global_load ... label: a load the compiler made conditional (usually
                          `ok ? p[i] : 0.f` in the source); such a load cannot be counted, its use waits for vmcnt(0);
  * drained loop heads -- a loop header whose first wait is `s_waitcnt vmcnt(0)` before the body has issued any load: the
                          loads in flight across the back edge (a prefetch) are all awaited there.
Neither is wrong by itself (rare paths, epilogues, third-party rocPRIM code); the list is where to look first.

And one pattern that IS wrong: `asm_mfma` -- a vector instruction inside an inline-assembly block
(`;;#ASMSTART` .. `;;#ASMEND`) close to a matrix instruction it shares a register with.  The compiler's hazard recogniser
inserts the wait states gfx950 needs between `v_mfma` and vector instructions only for instructions it can see; inline
assembly is opaque to it.  Flagged, conservatively (straight-line distance, every instruction = one wait state, `s_nop n`
= n + 1):
  * a `v_mfma` at most MFMA_BEFORE wait states BEFORE the asm instruction that writes a register the asm instruction reads
    or writes, or reads (A, B or C operand) a register the asm instruction writes;
  * a `v_mfma` at most MFMA_AFTER wait states AFTER it that reads a register the asm instruction writes.
And `asm_sgpr` (round 5: the ball query's inline-assembly v_addc read a lane mask that a vector compare had written too
recently -- duplicated hits): an inline-assembly vector instruction that reads a scalar register (or vcc) which a vector
instruction (v_cmp*, v_*_co_*, v_readlane, v_readfirstlane, v_div_scale*) wrote at most SGPR_BEFORE wait states earlier.
And `asm_dpp` (round 6): an inline-assembly DPP instruction less than DPP_BEFORE wait states behind the vector instruction that
wrote a register it reads (ball_grid.hip's v_min_i32_dpp reductions carry their own s_nop: this checks them under whatever
schedule the compiler chose around the block).
The script exits non-zero when one of any kind is found."""
import argparse
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = __file__.rsplit("/tools/", 1)[0]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def demangle(name):
    try:
        return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        return name


MFMA_BEFORE, MFMA_AFTER = 20, 6  # wait states: a 16-pass product's result -> vector read needs 19; vector write -> product read 2..4
SGPR_BEFORE = 4                  # wait states between a vector instruction's scalar result and a vector instruction that reads it
DPP_BEFORE = 2                   # wait states between a vector instruction's result and a DPP instruction that reads it


def sregs_of(operand):
    """{'s4', 's5', 'vcc'} named by one assembly operand (s4, s[4:5], vcc, vcc_lo ...; anything else: empty)."""
    if operand.startswith("vcc"):
        return {"vcc"}
    m = re.fullmatch(r"s(\d+)", operand)
    if m:
        return {f"s{int(m.group(1))}"}
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", operand)
    if m:
        return {f"s{i}" for i in range(int(m.group(1)), int(m.group(2)) + 1)}
    return set()


def valu_sgpr_writes(mn, ops):
    """scalar registers a vector instruction writes: v_cmp* (dst = operand 0, or vcc in the e32 form), carry-outs of v_*_co_*
    (operand 1), v_readlane / v_readfirstlane (operand 0), v_div_scale (operand 1)"""
    if mn.startswith("v_cmp"):
        return sregs_of(ops[0]) if ops and (ops[0].startswith("s") or ops[0].startswith("vcc")) else {"vcc"}
    if mn.startswith(("v_readlane", "v_readfirstlane")):
        return sregs_of(ops[0]) if ops else set()
    if "_co_" in mn or mn.startswith("v_div_scale"):
        return sregs_of(ops[1]) if len(ops) > 1 else set()
    return set()


def regs_of(operand):
    """{'v12', 'a3', ...} named by one assembly operand (v12, v[4:7], a[0:15]; anything else: empty)."""
    m = re.fullmatch(r"([va])(\d+)", operand)
    if m:
        return {f"{m.group(1)}{int(m.group(2))}"}
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", operand)
    if m:
        return {f"{m.group(1)}{i}" for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    return set()


def split_instr(t):
    """mnemonic, [operands] of one line of assembly (comments and modifiers such as `row_shr:1` dropped)."""
    t = t.split(";")[0].split("//")[0].strip()
    if not t or t.endswith(":") or t.startswith("."):
        return None, []
    parts = t.split(None, 1)
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    return parts[0], [o.split()[0] if o.split() else o for o in ops]


def asm_mfma_hazards(lines, start, end):
    """Inline-assembly vector instructions of lines[start:end] that share a register with a nearby v_mfma (see the header)."""
    instrs, in_asm = [], False  # (line number, mnemonic, writes, reads, inside inline asm, wait states it stands for)
    for k in range(start, end):
        t = lines[k].strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        mn, ops = split_instr(t)
        if mn is None:
            continue
        wait = 1
        if mn == "s_nop" and ops and ops[0].isdigit():
            wait = int(ops[0]) + 1
        regs = [regs_of(o) for o in ops]
        writes = regs[0] if regs and mn.startswith("v_") and not mn.startswith(("v_cmp", "v_readlane", "v_readfirstlane")) else set()
        reads = set().union(*regs[1:]) if len(regs) > 1 else set()
        if mn.startswith(("v_readlane", "v_readfirstlane", "v_cmp")):
            reads = set().union(*regs) if regs else set()
        if mn.startswith(("ds_", "global_", "buffer_", "flat_", "scratch_")):  # memory instructions: every register is read or a loaded destination
            writes, reads = set(), set().union(*regs) if regs else set()
        sread = set().union(*[sregs_of(o) for o in ops[1:]]) if len(ops) > 1 and mn.startswith("v_") else set()
        swrite = valu_sgpr_writes(mn, ops) if mn.startswith("v_") else set()
        if mn.startswith("v_") and "_co_" in mn and len(ops) > 1:
            sread -= sregs_of(ops[1])  # (operand 1 of a carry instruction is its carry-OUT)
        instrs.append((k + 1, mn, writes, reads, in_asm, wait, sread, swrite))
    found = []
    for i, (ln, mn, wr, rd, ia, _, srd, _sw) in enumerate(instrs):
        if not ia or not mn.startswith("v_"):
            continue
        dist = 0
        for j in range(i - 1, -1, -1):  # a scalar operand written by a vector instruction too recently
            l2, m2, _, _, _, wt, _, sw2 = instrs[j]
            dist += wt
            if dist > SGPR_BEFORE:
                break
            if srd & sw2:
                found.append(f"line {ln}: inline-asm `{mn}` reads {sorted(srd & sw2)[:3]} {dist} wait state(s) behind `{m2}` (line {l2})")
                break
        if "_dpp" in mn:  # asm_dpp: a DPP instruction reads another LANE's copy of a register a vector instruction
            # wrote: two wait states in between, which the hazard recogniser inserts for the compiler's own code only
            between = 0
            for j in range(i - 1, -1, -1):
                l2, m2, w2, _, _, wt, _, _ = instrs[j]
                if m2.startswith("v_") and (w2 & (rd | wr)):
                    if between < DPP_BEFORE:
                        found.append(f"line {ln}: inline-asm `{mn}` reads {sorted(w2 & (rd | wr))[:3]} {between} wait state(s) behind `{m2}` (line {l2}); {DPP_BEFORE} needed")
                    break
                between += wt
                if between >= DPP_BEFORE:
                    break
        dist = 0
        for j in range(i - 1, -1, -1):
            l2, m2, w2, r2, _, wt, _, _ = instrs[j]
            dist += wt
            if dist > MFMA_BEFORE:
                break
            if m2.startswith(("v_mfma", "v_smfmac")) and ((w2 & (rd | wr)) or (r2 & wr)):
                found.append(f"line {ln}: inline-asm `{mn}` {dist} wait state(s) behind `{m2}` (line {l2}) sharing {sorted((w2 & (rd | wr)) | (r2 & wr))[:4]}")
                break
        dist = 0
        for j in range(i + 1, len(instrs)):
            l2, m2, w2, r2, _, wt, _, _ = instrs[j]
            if m2.startswith(("v_mfma", "v_smfmac")) and (r2 & wr) and dist < MFMA_AFTER:
                found.append(f"line {ln}: inline-asm `{mn}` writes {sorted(r2 & wr)[:4]} {dist} wait state(s) ahead of `{m2}` (line {l2})")
                break
            dist += wt
            if dist >= MFMA_AFTER:
                break
    return found


def scan(asm):
    lines = asm.split("\n")
    kern, out = None, {}
    for k, l in enumerate(lines):
        m = re.match(r"^(_Z\S+):", l)
        if m:
            kern = m.group(1)
            out.setdefault(kern, {"pred": 0, "drain": [], "start": k})
        if kern is None:
            continue
        t = l.strip()
        if t.startswith("s_cbranch_execz"):
            loads, closed = 0, False
            for u in lines[k + 1:k + 14]:
                u = u.strip()
                loads += "global_load" in u or "buffer_load" in u
                if u.startswith(".LBB"):
                    closed = True
                    break
            if closed and loads == 1:
                out[kern]["pred"] += 1
        if "Loop Header" in l and l.startswith(".LBB"):
            for u in lines[k + 1:k + 14]:
                u = u.strip()
                if "global_load" in u or "buffer_load" in u:
                    break
                if u.startswith("s_waitcnt") and "vmcnt(0)" in u:
                    out[kern]["drain"].append(l.split(":")[0])
                    break
    starts = sorted((v["start"], kern) for kern, v in out.items())
    for i, (st, kern) in enumerate(starts):
        en = starts[i + 1][0] if i + 1 < len(starts) else len(lines)
        for k in range(st, en):
            if lines[k].strip().startswith(".Lfunc_end"):
                en = k
                break
        out[kern]["asm_mfma"] = asm_mfma_hazards(lines, st, en)
    meta = {}
    for m in re.finditer(r"\.agpr_count:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_count:\s+(\d+)",
                         asm, re.S):
        meta[m.group(2)] = (int(m.group(4)), int(m.group(1)), int(m.group(3)))
    return out, meta


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="*")
    ap.add_argument("--out")
    a = ap.parse_args()
    files = a.files or sorted(glob.glob(os.path.join(ROOT, "pointasnl_amd", "csrc", "*.hip")))
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for f in files:
            s = os.path.join(tmp, os.path.basename(f) + ".s")
            r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S",
                                "--cuda-device-only", "-I", os.path.join(ROOT, "include"), f, "-o", s], capture_output=True, text=True)
            if r.returncode != 0:
                print(f"{f}: hipcc failed\n{r.stderr[-2000:]}", file=sys.stderr)
                continue
            res, meta = scan(open(s).read())
            for kern, v in res.items():
                if kern not in meta:
                    continue  # a device function, not a kernel
                vg, ag, sc = meta[kern]
                rows.append((os.path.basename(f), demangle(kern), vg, ag, sc, v["pred"], len(v["drain"]), v["asm_mfma"]))
    rows.sort(key=lambda r: (-(r[5] + 4 * r[6]), r[0], r[1]))
    hazards = [(f, k, h) for f, k, _, _, _, _, _, hz in rows for h in hz]
    text = [f"inline-assembly vector instructions sharing a register with a matrix instruction within {MFMA_BEFORE} wait states before / "
            f"{MFMA_AFTER} after, or reading a scalar register a vector instruction wrote within {SGPR_BEFORE}: {len(hazards)}"
            + (" (clean)" if not hazards else "")]
    for f, k, h in hazards:
        text.append(f"  HAZARD {f}  {k[:120]}\n         {h}")
    text.append("")
    text.append("file            vgpr(total) agpr scratch predicated_loads drained_loop_heads asm_mfma  kernel")
    for f, k, vg, ag, sc, pr, dr, hz in rows:
        if "rocprim" in k:
            k = "rocprim::" + k.split("rocprim::")[-1][:60] + " (third party)"
        text.append(f"{f:<16}{vg:>10} {ag:>5} {sc:>7} {pr:>16} {dr:>18} {len(hz):>8}  {k[:150]}")
    text = "\n".join(text)
    print(text)
    if a.out:
        open(a.out, "w").write(text + "\n")
    sys.exit(1 if hazards else 0)


if __name__ == "__main__":
    main()

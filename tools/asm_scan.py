"""Static scan of the gfx950 assembly of every kernel in pointasnl_amd/csrc (no GPU needed: hipcc cross-compiles).

  python tools/asm_scan.py [--out profiles/<tag>_asm_scan.txt] [file.hip ...]

Per kernel: registers / scratch, and the two patterns DESIGN.md 6 describes ("waits across a loop's back edge", "loads behind
a select"):
  * predicated loads   -- `s_cbranch_execz` ... ONE global_load ... label: a load the compiler made conditional (usually
                          `ok ? p[i] : 0.f` in the source); such a load cannot be counted, its use waits for vmcnt(0);
  * drained loop heads -- a loop header whose first wait is `s_waitcnt vmcnt(0)` before the body has issued any load: the
                          loads in flight across the back edge (a prefetch) are all awaited there.
Neither is wrong by itself (rare paths, epilogues, third-party rocPRIM code); the list is where to look first."""
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


def scan(asm):
    lines = asm.split("\n")
    kern, out = None, {}
    for k, l in enumerate(lines):
        m = re.match(r"^(_Z\S+):", l)
        if m:
            kern = m.group(1)
            out.setdefault(kern, {"pred": 0, "drain": []})
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
                rows.append((os.path.basename(f), demangle(kern), vg, ag, sc, v["pred"], len(v["drain"])))
    rows.sort(key=lambda r: (-(r[5] + 4 * r[6]), r[0], r[1]))
    text = ["file            vgpr(total) agpr scratch predicated_loads drained_loop_heads  kernel"]
    for f, k, vg, ag, sc, pr, dr in rows:
        if "rocprim" in k:
            k = "rocprim::" + k.split("rocprim::")[-1][:60] + " (third party)"
        text.append(f"{f:<16}{vg:>10} {ag:>5} {sc:>7} {pr:>16} {dr:>18}  {k[:150]}")
    text = "\n".join(text)
    print(text)
    if a.out:
        open(a.out, "w").write(text + "\n")


if __name__ == "__main__":
    main()
